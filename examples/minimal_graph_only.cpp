// Graph-only problem in double precision: fit (a,b) of y = a cos(bx) + b sin(ax) to 512 samples (energy:
// opt_amd/energies/curveFitting.t).  Counterpart of the reference's tests/minimal_graph_only -- its one known-answer test:
// starting from (99.7, 101.6) the solver must recover the generator parameters (100, 102).
#include "common.h"
#include <cmath>

int main(int argc, char** argv) {
    const std::string energy = argc > 1 ? argv[1] : "opt_amd/energies/curveFitting.t";
    const int n = 512;
    const double a = 100.0, b = 102.0;
    std::vector<double> data(2 * n);
    for (int i = 0; i < n; ++i) {
        const double x = (double)(float)i * 2.0 * 3.141592653589 / n;
        data[2 * i] = x; data[2 * i + 1] = a * std::cos(b * x) + b * std::sin(a * x);
    }
    std::vector<double> init = {(double)99.7f, (double)101.6f};
    std::vector<int> sampleIdx(n), paramIdx(n, 0);
    for (int i = 0; i < n; ++i) sampleIdx[i] = i;
    DeviceBuffer<double> dData(data), dUnknown(init);
    DeviceBuffer<int> dSample(sampleIdx), dParam(paramIdx);

    Opt_InitializationParameters param = {};
    param.doublePrecision = 1;
    param.verbosityLevel = 1;
    Opt_State* state = Opt_NewState(param);
    if (!state) return 2;
    Opt_Problem* problem = Opt_ProblemDefine(state, energy.c_str(), "gaussNewtonGPU");
    unsigned int dims[] = {(unsigned)n, 1};
    Opt_Plan* plan = Opt_ProblemPlan(state, problem, dims);
    if (!plan) { fprintf(stderr, "plan failed\n"); return 3; }
    int edgeCount = n;                                                       // read on the host by the library
    void* problem_data[] = {dUnknown.ptr, dData.ptr, &edgeCount, dSample.ptr, dParam.ptr};
    Opt_ProblemSolve(state, plan, problem_data);
    Opt_PlanFree(state, plan);
    Opt_ProblemDelete(state, problem);

    const std::vector<double> res = dUnknown.download();
    std::cout << std::setprecision(12) << "Init " << init[0] << ", " << init[1] << std::endl << "Result " << res[0] << ", " << res[1] << std::endl
              << "Goal " << a << ", " << b << std::endl;
    return (std::abs(res[0] - a) < 1e-6 && std::abs(res[1] - b) < 1e-6) ? 0 : 1;
}
