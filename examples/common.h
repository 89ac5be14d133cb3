// Shared helpers of the C++ example callers.  These programs are the C/C++ side of the drop-in boundary: they include
// include/Opt.h, link opt_amd/lib/libOpt.so and allocate their buffers with the HIP runtime -- the way the reference's
// tests/ and examples/ programs use libOpt.a with CUDA (tests/minimal/main.cpp, examples/shared/OptSolver.h:40-97,
// OptUtils.h:47-64, SolverIteration.h:28-86).  Nothing here is copied from the reference; the printed formats
// ("cost: a -> b", "===name=== / **Final Costs**", the results CSV header) are reproduced because harness scripts
// grep them (scripts/print_all_costs.py).
#pragma once
extern "C" {
#include "Opt.h"
}
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#define EX_HIP(call)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
    } while (0)

template <class T>
struct DeviceBuffer {   // owning device array, uploaded from / downloaded to a std::vector
    T* ptr = nullptr; size_t n = 0;
    DeviceBuffer() {}
    explicit DeviceBuffer(size_t count) { alloc(count); }
    explicit DeviceBuffer(const std::vector<T>& h) { alloc(h.size()); upload(h); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { if (ptr) (void)hipFree(ptr); }
    void alloc(size_t count) { n = count; EX_HIP(hipMalloc((void**)&ptr, n * sizeof(T))); EX_HIP(hipMemset(ptr, 0, n * sizeof(T))); }
    void upload(const std::vector<T>& h) { EX_HIP(hipMemcpy(ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
    std::vector<T> download() const { std::vector<T> h(n); EX_HIP(hipMemcpy(h.data(), ptr, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }
};

// OPT_EXAMPLE_DUMP=<directory>: the example writes the host arrays it hands to the solver as raw little-endian files <directory>/<name>.bin, so that a test can
// run the CPU oracle on exactly the caller's inputs (tests/test_cpp_callers_gpu.py) instead of re-deriving them.
template <class T>
inline void dumpInput(const std::string& name, const std::vector<T>& v) {
    const char* dir = getenv("OPT_EXAMPLE_DUMP");
    if (!dir) return;
    std::ofstream f(std::string(dir) + "/" + name + ".bin", std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

struct SolverIteration { double cost; double timeInMS; };

// Init + Step loop with a (cost, ms) record per outer iteration (what OptUtils.h's launchProfiledSolve collects).
inline void profiledSolve(Opt_State* state, Opt_Plan* plan, void** params, std::vector<SolverIteration>& iters) {
    using clock = std::chrono::steady_clock;
    EX_HIP(hipDeviceSynchronize());
    auto t0 = clock::now();
    Opt_ProblemInit(state, plan, params);
    EX_HIP(hipDeviceSynchronize());
    iters.push_back({Opt_ProblemCurrentCost(state, plan), std::chrono::duration<double, std::milli>(clock::now() - t0).count()});
    for (;;) {
        t0 = clock::now();
        const int more = Opt_ProblemStep(state, plan, params);
        EX_HIP(hipDeviceSynchronize());
        if (!more) break;
        iters.push_back({Opt_ProblemCurrentCost(state, plan), std::chrono::duration<double, std::milli>(clock::now() - t0).count()});
    }
}

inline void saveSolverResults(const std::string& path, const std::vector<SolverIteration>& gn, const std::vector<SolverIteration>& lm, bool dbl) {
    std::ofstream f(path);
    f << std::scientific << std::setprecision(20);
    const std::string sfx = dbl ? " (double)" : " (float)";
    f << "Iter, Opt(GN) Error" << sfx << ",  Opt(LM) Error" << sfx << ", Opt(GN) Iter Time(ms)" << sfx << ", Opt(LM) Iter Time(ms)" << sfx
      << ", Total Opt(GN) Time(ms)" << sfx << ", Total Opt(LM) Time(ms)" << sfx << std::endl;
    double sg = 0, sl = 0;
    auto at = [](const std::vector<SolverIteration>& v, size_t i) { return v.empty() ? SolverIteration{0, 0} : v[std::min(i, v.size() - 1)]; };
    for (size_t i = 0; i < std::max(gn.size(), lm.size()); ++i) {
        const double tg = i < gn.size() ? gn[i].timeInMS : 0, tl = i < lm.size() ? lm[i].timeInMS : 0;
        sg += tg; sl += tl;
        f << i << ", " << at(gn, i).cost << ", " << at(lm, i).cost << ", " << tg << ", " << tl << ", " << sg << ", " << sl << std::endl;
    }
}

inline void reportFinalCosts(const std::string& name, bool useGN, double gnCost, bool useLM, double lmCost) {
    std::cout << "===" << name << "===" << std::endl << "**Final Costs**" << std::endl << "Opt GN,Opt LM,CERES" << std::endl;
    std::cout << std::scientific << std::setprecision(20);
    if (useGN) std::cout << gnCost;
    std::cout << ",";
    if (useLM) std::cout << lmCost;
    std::cout << "," << std::endl;
}
