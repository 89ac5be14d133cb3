// Smallest image problem through the C API: smooth a random 512x512 image (energy: opt_amd/energies/laplacian.t).
// Counterpart of the reference's tests/minimal (random target, unknown initialised to the target, GN defaults 10 x 10).
#include "common.h"

int main(int argc, char** argv) {
    const std::string energy = argc > 1 ? argv[1] : "opt_amd/energies/laplacian.t";
    const unsigned dim = 512;
    std::vector<float> target(dim * dim);
    unsigned s = 12345u;
    for (auto& v : target) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / (float)(1u << 24); }
    DeviceBuffer<float> dTarget(target), dUnknown(target);

    Opt_InitializationParameters param = {};
    param.verbosityLevel = 1;
    param.collectPerKernelTimingInfo = 1;
    Opt_State* state = Opt_NewState(param);
    if (!state) return 2;
    Opt_Problem* problem = Opt_ProblemDefine(state, energy.c_str(), "gaussNewtonGPU");
    unsigned int dims[] = {dim, dim};
    Opt_Plan* plan = Opt_ProblemPlan(state, problem, dims);
    if (!plan) { fprintf(stderr, "plan failed\n"); return 3; }
    void* problem_data[] = {dUnknown.ptr, dTarget.ptr};
    Opt_ProblemInit(state, plan, problem_data);
    const double c0 = Opt_ProblemCurrentCost(state, plan);
    while (Opt_ProblemStep(state, plan, problem_data)) {}
    const double c1 = Opt_ProblemCurrentCost(state, plan);
    Opt_PlanFree(state, plan);
    Opt_ProblemDelete(state, problem);

    const std::vector<float> out = dUnknown.download();
    double roughIn = 0, roughOut = 0;
    for (unsigned y = 0; y < dim; ++y) for (unsigned x = 0; x + 1 < dim; ++x) {
        roughIn += std::abs(target[y * dim + x] - target[y * dim + x + 1]); roughOut += std::abs(out[y * dim + x] - out[y * dim + x + 1]);
    }
    printf("Init cost %g  Result cost %g  roughness %g -> %g\n", c0, c1, roughIn, roughOut);
    return (c1 < c0 && roughOut < roughIn) ? 0 : 1;
}
