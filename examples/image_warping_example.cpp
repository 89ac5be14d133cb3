// image_warping through the C API, driven like the reference example (examples/image_warping/src/main.cpp:98-139,
// CombinedSolver.h:104-207, examples/shared/CombinedSolverBase.h:98-119): the constraint image is ramped towards the
// target marker positions over `passes` outer passes, each pass one solve of nonLinearIter x linearIter; GN and LM
// solvers run on identical inputs; per-step (cost, ms) go to results_<float|double>.csv and the final costs are printed
// in the block the reference's scripts grep.  The image is procedural (no PNG decoder here): W x W grid, border pinned,
// the nine cat512 markers scaled to the image size.
//   usage: image_warping_example [size=512] [passes=19] [nonLinearIter=8] [linearIter=400] [energy.t] [double=0]
#include "common.h"
#include <cmath>

static const int kMarkers[9][4] = {{30, 132, 59, 44}, {229, 51, 157, 91}, {430, 124, 379, 42}, {281, 369, 326, 323}, {197, 407, 163, 418},
                                   {64, 386, 26, 300}, {311, 168, 253, 182}, {89, 228, 56, 255}, {92, 192, 84, 192}};

template <class T>
struct Warp {
    unsigned W, H;
    std::vector<T> urshape, mask;
    DeviceBuffer<T> dOffset, dAngle, dUrshape, dConstraints, dMask;
    float wFitSqrt = std::sqrt(100.0f), wRegSqrt = std::sqrt(0.01f);
    Warp(unsigned w, unsigned h) : W(w), H(h), urshape(2 * w * h), mask(w * h, T(0)), dOffset(2 * w * h), dAngle(w * h), dUrshape(2 * w * h), dConstraints(2 * w * h), dMask(w * h) {
        for (unsigned y = 0; y < H; ++y) for (unsigned x = 0; x < W; ++x) { urshape[2 * (y * W + x)] = (T)x; urshape[2 * (y * W + x) + 1] = (T)y; }
        dUrshape.upload(urshape); dMask.upload(mask);
    }
    void reset() { dOffset.upload(urshape); EX_HIP(hipMemset(dAngle.ptr, 0, dAngle.n * sizeof(T))); setConstraints(1.0f); }
    void setConstraints(float alpha) {   // markers move from their source (alpha = 0) to their target (alpha = 1)
        std::vector<T> c(2 * W * H, T(-1));
        auto pin = [&](unsigned x, unsigned y, T tx, T ty) { c[2 * (y * W + x)] = tx; c[2 * (y * W + x) + 1] = ty; };
        for (unsigned x = 0; x < W; ++x) { pin(x, 0, (T)x, 0); pin(x, H - 1, (T)x, (T)(H - 1)); }
        for (unsigned y = 0; y < H; ++y) { pin(0, y, 0, (T)y); pin(W - 1, y, (T)(W - 1), (T)y); }
        for (auto& m : kMarkers) {
            const unsigned x = m[0] * W / 512, y = m[1] * H / 512; const float tx = (float)(m[2] * W / 512), ty = (float)(m[3] * H / 512);
            pin(x, y, (T)((1.0f - alpha) * (float)x + alpha * tx), (T)((1.0f - alpha) * (float)y + alpha * ty));
        }
        dConstraints.upload(c);
    }
};

template <class T>
int run(unsigned size, int passes, int nonLinearIter, int linearIter, const std::string& energy, bool dbl) {
    Warp<T> warp(size, size);
    Opt_InitializationParameters ip = {};
    ip.doublePrecision = dbl ? 1 : 0;
    Opt_State* state = Opt_NewState(ip);
    if (!state) return 2;
    unsigned int dims[] = {size, size};
    std::vector<SolverIteration> iters[2];
    double finalCost[2] = {0, 0};
    const char* kinds[2] = {"gaussNewtonGPU", "LMGPU"};
    const char* names[2] = {"Opt(GN)", "Opt(LM)"};
    for (int k = 0; k < 2; ++k) {
        Opt_Problem* problem = Opt_ProblemDefine(state, energy.c_str(), kinds[k]);
        Opt_Plan* plan = Opt_ProblemPlan(state, problem, dims);
        if (!plan) return 3;
        Opt_SetSolverParameter(state, plan, "nIterations", &nonLinearIter);
        Opt_SetSolverParameter(state, plan, "lIterations", &linearIter);
        void* params[] = {warp.dOffset.ptr, warp.dAngle.ptr, warp.dUrshape.ptr, warp.dConstraints.ptr, warp.dMask.ptr, &warp.wFitSqrt, &warp.wRegSqrt};
        warp.reset();
        for (int i = 0; i < passes; ++i) {
            std::cout << "//////////// ITERATION" << i << "  (" << names[k] << ") ///////////////" << std::endl;
            warp.setConstraints((float)(i + 1) / (float)passes);
            const size_t before = iters[k].size();
            profiledSolve(state, plan, params, iters[k]);
            for (size_t j = before + 1; j < iters[k].size(); ++j) printf("cost: %f -> %f\n", iters[k][j - 1].cost, iters[k][j].cost);
        }
        finalCost[k] = Opt_ProblemCurrentCost(state, plan);
        Opt_PlanFree(state, plan);
        Opt_ProblemDelete(state, problem);
    }
    saveSolverResults(std::string("results_") + (dbl ? "double" : "float") + ".csv", iters[0], iters[1], dbl);
    reportFinalCosts("Image Warping", true, finalCost[0], true, finalCost[1]);
    double ms[2] = {0, 0};
    for (int k = 0; k < 2; ++k) for (auto& it : iters[k]) ms[k] += it.timeInMS;
    std::cout << std::fixed << std::setprecision(2) << "total solver time: GN " << ms[0] << " ms, LM " << ms[1] << " ms" << std::endl;
    return (std::isfinite(finalCost[0]) && std::isfinite(finalCost[1])) ? 0 : 1;
}

int main(int argc, char** argv) {
    const unsigned size = argc > 1 ? atoi(argv[1]) : 512;
    const int passes = argc > 2 ? atoi(argv[2]) : 19, nl = argc > 3 ? atoi(argv[3]) : 8, li = argc > 4 ? atoi(argv[4]) : 400;
    const std::string energy = argc > 5 ? argv[5] : "opt_amd/energies/image_warping.t";
    const bool dbl = argc > 6 && atoi(argv[6]) != 0;
    return dbl ? run<double>(size, passes, nl, li, energy, true) : run<float>(size, passes, nl, li, energy, false);
}
