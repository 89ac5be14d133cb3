// Plan create / free in a loop: device memory must come back (counterpart of the reference's tests/create_delete_cycle).
#include "common.h"

int main(int argc, char** argv) {
    const std::string energy = argc > 1 ? argv[1] : "opt_amd/energies/laplacian.t";
    const int cycles = argc > 2 ? atoi(argv[2]) : 1000;
    Opt_InitializationParameters param = {};
    Opt_State* state = Opt_NewState(param);
    if (!state) return 2;
    unsigned int dims[] = {512, 512};
    size_t free0 = 0, free1 = 0, total = 0;
    for (int i = -1; i < cycles; ++i) {
        if (i == 0) EX_HIP(hipMemGetInfo(&free0, &total));   // after one warm-up cycle: code objects, queues and the runtime's pools exist
        Opt_Problem* problem = Opt_ProblemDefine(state, energy.c_str(), "gaussNewtonGPU");
        Opt_Plan* plan = Opt_ProblemPlan(state, problem, dims);
        if (!plan) return 3;
        Opt_PlanFree(state, plan);
        Opt_ProblemDelete(state, problem);
    }
    EX_HIP(hipMemGetInfo(&free1, &total));
    printf("%d plan create/free cycles, device memory delta %ld bytes\n", cycles, (long)free0 - (long)free1);
    return ((long)free0 - (long)free1 < (64l << 20)) ? 0 : 1;
}
