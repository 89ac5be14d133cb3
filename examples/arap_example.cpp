// arap_mesh_deformation through the C API, driven like the reference example (examples/arap_mesh_deformation/src/
// main.cpp:75-104, CombinedSolver.h:62-164): per-vertex Offset / Angle unknowns, directed half-edges grouped by head vertex
// (examples/shared/OptGraph.h:64-76), handle constraints ramped towards their targets over `passes` passes of
// nonLinearIter x linearIter.  The mesh is a procedural triangulated grid (the reference's mesh files are not shipped).
//   usage: arap_example [nx=200] [ny=200] [passes=10] [nonLinearIter=20] [linearIter=100] [energy.t] [double=0]
#include "common.h"
#include <cmath>
#include <limits>

template <class T>
int run(int nx, int ny, int passes, int nonLinearIter, int linearIter, const std::string& energy, bool dbl) {
    const int N = nx * ny;
    std::vector<T> rest(3 * N);
    for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) { const int v = y * nx + x; rest[3 * v] = (T)(0.01f * x); rest[3 * v + 1] = (T)(0.01f * y); rest[3 * v + 2] = (T)(0.02f * std::sin(0.1f * x) * std::cos(0.13f * y)); }
    std::vector<int> head, tail;
    const int nb[6][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {-1, -1}};
    for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) for (auto& d : nb) {
        const int tx = x + d[0], ty = y + d[1];
        if (tx >= 0 && tx < nx && ty >= 0 && ty < ny) { head.push_back(y * nx + x); tail.push_back(ty * nx + tx); }
    }
    int edgeCount = (int)head.size();
    const T ninf = -std::numeric_limits<T>::infinity();
    auto constraints = [&](float alpha) {          // left column pinned, right column pulled to its target
        std::vector<T> c(3 * N, ninf);
        for (int y = 0; y < ny; ++y) {
            const int l = y * nx, r = y * nx + nx - 1;
            for (int k = 0; k < 3; ++k) c[3 * l + k] = rest[3 * l + k];
            c[3 * r] = rest[3 * r]; c[3 * r + 1] = rest[3 * r + 1] + (T)(alpha * 0.05f * ny * 0.01f); c[3 * r + 2] = rest[3 * r + 2] + (T)(alpha * 0.1f * nx * 0.01f);
        }
        return c;
    };
    dumpInput("arap_rest", rest); dumpInput("arap_head", head); dumpInput("arap_tail", tail);
    DeviceBuffer<T> dOffset(rest), dAngle(3 * (size_t)N), dRest(rest), dCons(3 * (size_t)N);
    DeviceBuffer<int> dHead(head), dTail(tail);
    float wFitSqrt = std::sqrt(4.0f), wRegSqrt = std::sqrt(1.0f);
    Opt_InitializationParameters ip = {};
    ip.doublePrecision = dbl ? 1 : 0;
    Opt_State* state = Opt_NewState(ip);
    if (!state) return 2;
    unsigned int dims[] = {(unsigned)N};
    Opt_Problem* problem = Opt_ProblemDefine(state, energy.c_str(), "gaussNewtonGPU");
    Opt_Plan* plan = Opt_ProblemPlan(state, problem, dims);
    if (!plan) return 3;
    Opt_SetSolverParameter(state, plan, "nIterations", &nonLinearIter);
    Opt_SetSolverParameter(state, plan, "lIterations", &linearIter);
    void* params[] = {&wFitSqrt, &wRegSqrt, dOffset.ptr, dAngle.ptr, dRest.ptr, dCons.ptr, &edgeCount, dHead.ptr, dTail.ptr};
    std::vector<SolverIteration> iters, none;
    for (int i = 0; i < passes; ++i) {
        std::cout << "//////////// ITERATION" << i << "  (Opt(GN)) ///////////////" << std::endl;
        const std::vector<T> c = constraints((float)(i + 1) / (float)passes);
        dumpInput("arap_constraints_" + std::to_string(i), c);
        dCons.upload(c);
        profiledSolve(state, plan, params, iters);
    }
    const double finalCost = Opt_ProblemCurrentCost(state, plan);
    Opt_PlanFree(state, plan); Opt_ProblemDelete(state, problem);
    const std::vector<T> off = dOffset.download();
    const int r = (ny / 2) * nx + nx - 1;
    printf("%d vertices, %d half-edges; handle vertex %d moved to (%.4f, %.4f, %.4f)\n", N, edgeCount, r, (double)off[3 * r], (double)off[3 * r + 1], (double)off[3 * r + 2]);
    saveSolverResults(std::string("results_") + (dbl ? "double" : "float") + ".csv", iters, none, dbl);
    reportFinalCosts("Mesh Deformation ARAP", true, finalCost, false, 0.0);
    double ms = 0; for (auto& it : iters) ms += it.timeInMS;
    std::cout << std::fixed << std::setprecision(2) << "total solver time: " << ms << " ms" << std::endl;
    const double targetZ = (double)rest[3 * r + 2] + 0.1f * nx * 0.01f;
    return (std::isfinite(finalCost) && std::abs((double)off[3 * r + 2] - targetZ) < 0.2 * std::abs(0.1f * nx * 0.01f)) ? 0 : 1;
}

int main(int argc, char** argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 200, ny = argc > 2 ? atoi(argv[2]) : 200, passes = argc > 3 ? atoi(argv[3]) : 10;
    const int nonLinearIter = argc > 4 ? atoi(argv[4]) : 20, linearIter = argc > 5 ? atoi(argv[5]) : 100;
    const std::string energy = argc > 6 ? argv[6] : "opt_amd/energies/arap_mesh_deformation.t";
    const bool dbl = argc > 7 && atoi(argv[7]) != 0;
    return dbl ? run<double>(nx, ny, passes, nonLinearIter, linearIter, energy, true) : run<float>(nx, ny, passes, nonLinearIter, linearIter, energy, false);
}
