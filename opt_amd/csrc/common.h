// Device primitives and launch plumbing shared by every kernel set (gfx950 / wave64 only).
//
// Replaces the reference's API/src/util.t: warpReduce (:612-623, 5-step shfl over 32 lanes) becomes a
// 6-step wave64 reduction; "lane 0 does one atomicAdd on a global scalar" (solverGPUGaussNewton.t:312-317)
// becomes "one partial per workgroup, summed in fixed order by the consumer" -- deterministic, no
// same-address atomics, no memsets between kernels; makeGPULauncher's event pairs (util.t:800-843) become
// the KernelTimer below.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <vector>

// A/B switches of questions that are settled (DESIGN.md section 3 says how): a product build does not read them -- the default is compiled in, the other arm is
// reachable only in a development build (-DOPT_AMD_DEV_SWITCHES: opt_amd/build.py build_variant("dev", ["OPT_AMD_DEV_SWITCHES"])).
inline int devSwitch(const char* name, int dflt) {
#ifdef OPT_AMD_DEV_SWITCHES
    if (const char* e = getenv(name)) return atoi(e);
#endif
    (void)name;
    return dflt;
}

#define HIP_CHECK(call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            /* reference util.t:739-753: print and exit with the error code */                          \
            fprintf(stderr, "HIP reported error %d: %s\nIn call: %s\nIn file: %s:%d\n", (int)e_,        \
                    hipGetErrorString(e_), #call, __FILE__, __LINE__);                                    \
            exit((int)e_);                                                                                \
        }                                                                                                 \
    } while (0)

namespace optamd {

constexpr int kWave = 64;
constexpr int kBlock = 256;          // threads per workgroup of the streaming kernels (4 waves)
constexpr int kMaxPartials = 4096;   // per-workgroup partial sums a reduction may produce (ARAP: 2048 vertex-pass + 2048 edge-pass workgroups)

// A grid-wide sum in flight: the producer kernel writes one double per workgroup into `partials`
// (count `n`, known on the host at launch), the consumer kernels sum them in index order.
struct Reduction {
    double* partials = nullptr;   // device, kMaxPartials doubles
    int n = 0;                    // how many the last producer wrote
    bool hostVisible = false;     // partials live in pinned host memory (zero-copy): a sum only the HOST consumes (LM's Q) needs no copy kernel
};

// A double delivered to the polling host as two self-validating 8-byte words {tag | low half, tag | high half} in pinned memory (slot 2i, 2i+1 of a
// host-visible Reduction): each word arrives whole, so the host needs neither an event in the stream nor a fence -- it spins until both words carry
// the launch's tag.  (The peer communicator's mailbox uses the same word format between GPUs.)
__device__ __forceinline__ void storeTaggedPartial(double* partials, int i, double v, unsigned tag) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)tag << 32;
    unsigned long long* w = reinterpret_cast<unsigned long long*>(partials) + 2 * (long)i;
    __hip_atomic_store(w, t | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(w + 1, t | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- wave64 / workgroup reductions -------------------------------------------------------------------
__device__ __forceinline__ double waveReduceSum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;   // valid in lane 0
}

// Sum `v` over the workgroup; the result is valid in thread 0.  `scratch` holds blockDim/64 doubles.
__device__ __forceinline__ double blockReduceSum(double v, double* scratch) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    v = waveReduceSum(v);
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x * blockDim.y + kWave - 1) / kWave;
        for (int i = 0; i < nw; ++i) t += scratch[i];
    }
    __syncthreads();
    return t;
}

// Every thread of the workgroup obtains sum(partials[0..n)) -- same value, same order, in every
// workgroup of every kernel: the deterministic replacement for reading a scalar that N/32 atomics built.
__device__ __forceinline__ double sumPartials(const double* __restrict__ partials, int n, double* scratch /* >= blockDim/64 + 1 */) {
    double t = 0;
    const int tid = threadIdx.x + threadIdx.y * blockDim.x, nt = blockDim.x * blockDim.y;
    for (int i = tid; i < n; i += nt) t += partials[i];
    const int lane = tid & (kWave - 1), wave = tid >> 6, nw = (nt + kWave - 1) / kWave;
    t = waveReduceSum(t);
    if (lane == 0) scratch[wave] = t;
    __syncthreads();
    if (tid == 0) { double s = 0; for (int i = 0; i < nw; ++i) s += scratch[i]; scratch[nw] = s; }
    __syncthreads();
    const double r = scratch[nw];
    __syncthreads();
    return r;
}

// K sums at once: the same per-array arithmetic as K calls of sumPartials / blockReduceSum (same order, same bits), but the K
// global loads are in flight together and the workgroup synchronises once per phase instead of once per array.  The prologue of
// a per-iteration kernel reads partials another kernel just wrote (an L2 miss of 1-2 us each), so this is worth ~6 us per launch.
// scratch: K * (blockDim/64 + 1) doubles.
template <int K>
__device__ __forceinline__ void sumPartialsN(const double* const (&partials)[K], const int (&n)[K], double* scratch, double (&out)[K]) {
    const int tid = threadIdx.x + threadIdx.y * blockDim.x, nt = blockDim.x * blockDim.y;
    const int lane = tid & (kWave - 1), wave = tid >> 6, nw = (nt + kWave - 1) / kWave;
    double t[K];
    // every array's first element is requested before anything is added: written as K loops the compiler waits for each array's load before it requests the next one --
    // K dependent memory round trips in the prologue of every iteration kernel (ISA of iw_pcgIter2, round 4) -- although a workgroup rarely needs a second pass
#pragma unroll
    for (int k = 0; k < K; ++k) t[k] = tid < n[k] ? partials[k][tid] : 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) for (int i = tid + nt; i < n[k]; i += nt) t[k] += partials[k][i];
#pragma unroll
    for (int k = 0; k < K; ++k) { t[k] = waveReduceSum(t[k]); if (lane == 0) scratch[k * (nw + 1) + wave] = t[k]; }
    __syncthreads();
    if (tid < K) { double s = 0; for (int i = 0; i < nw; ++i) s += scratch[tid * (nw + 1) + i]; scratch[tid * (nw + 1) + nw] = s; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = scratch[k * (nw + 1) + nw];
    __syncthreads();
}
// K workgroup sums at once; results valid in thread 0.  scratch: K * blockDim/64 doubles.
template <int K>
__device__ __forceinline__ void blockReduceSumN(double (&v)[K], double* scratch) {
    const int tid = threadIdx.x + threadIdx.y * blockDim.x, nt = blockDim.x * blockDim.y;
    const int lane = tid & (kWave - 1), wave = tid >> 6, nw = (nt + kWave - 1) / kWave;
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k] = waveReduceSum(v[k]); if (lane == 0) scratch[k * nw + wave] = v[k]; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) { double t = 0; for (int i = 0; i < nw; ++i) t += scratch[k * nw + i]; v[k] = t; }
    }
    __syncthreads();
}

// ---- whole-wave shifts --------------------------------------------------------------------------------------
// DPP whole-wave shifts (gfx9 family): wave_shr:1 gives lane i the value of lane i-1, wave_shl:1 of lane i+1;
// lanes shifted in from outside the wave read 0 (bound_ctrl).  One v_mov_b32_dpp per 32-bit word, no LDS.
template <bool RIGHT> __device__ __forceinline__ int dppShift(int v) {
    return RIGHT ? __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);
}
template <bool RIGHT> __device__ __forceinline__ float dppShift(float v) { return __int_as_float(dppShift<RIGHT>(__float_as_int(v))); }
template <bool RIGHT> __device__ __forceinline__ double dppShift(double v) {
    const int lo = dppShift<RIGHT>(__double2loint(v)), hi = dppShift<RIGHT>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// The K sums of a POSTED all-reduce (OptAmd_MailRef, include/OptAmd.h): every rank's contribution arrives in this rank's mailbox as tagged 8-byte words;
// thread (source rank r, word w) polls its word until it carries the tag, then K threads add the contributions in rank order -- the same bits on every
// rank and in every workgroup.  A poll that outlasts the time-out raises the communicator's error flag and yields NaN.  scratch: >= world * 2K unsigned +
// K doubles (the caller's reduction scratch is reused).  K <= 8, world * 2K <= blockDim.
struct MailRefDev { const unsigned long long* words; int world, stride; unsigned tag; long long timeoutTicks; int* errFlag; };
template <int K>
__device__ __forceinline__ void pollMailSums(const MailRefDev& M, double* scratch, double (&out)[K]) {
    unsigned* halves = reinterpret_cast<unsigned*>(scratch + K + 1);
    int* bad = reinterpret_cast<int*>(scratch + K);
    const int tid = threadIdx.x + threadIdx.y * blockDim.x, nw = 2 * K;
    if (tid == 0) *bad = 0;
    __syncthreads();
    if (tid < M.world * nw) {
        const int r = tid / nw, w = tid % nw;
        const unsigned long long* src = M.words + (long)r * M.stride + w;
        unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(v >> 32) != M.tag) {
            const long long t0 = wall_clock64();
            while ((unsigned)((v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >> 32) != M.tag) {
                __builtin_amdgcn_s_sleep(4);      // ~0.1 us between looks: one wave of every workgroup of the launch polls the same few hundred bytes while the peers' stores are landing
                if (wall_clock64() - t0 > M.timeoutTicks) { *bad = 1; break; }
            }
        }
        halves[tid] = (unsigned)v;
    }
    __syncthreads();
    if (tid < K) {
        double t = 0;
        for (int r = 0; r < M.world; ++r)
            t += __longlong_as_double((long long)(((unsigned long long)halves[r * nw + 2 * tid + 1] << 32) | halves[r * nw + 2 * tid]));
        if (*bad) { t = __longlong_as_double(0x7ff8000000000000ll); if (tid == 0 && blockIdx.x == 0) __hip_atomic_store(M.errFlag, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        scratch[tid] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = scratch[k];
    __syncthreads();
}

// The producer side (OptAmd_MailPost): called by every workgroup of a kernel after thread 0 has stored the workgroup's K partial sums partials[k][blockIdx.x].
// The workgroup that draws the last ticket sums the gridDim.x partials of each array with its first 256 threads -- the arithmetic of the communicator's own
// post kernel (k_mailPost), hence the same bits -- and stores the tagged words into every rank's mailbox.  blockDim >= 256.
struct MailPostDev { unsigned long long* dst[16]; int world; unsigned tag; unsigned* ticket; };
template <int K>
__device__ __forceinline__ void postMailSums(const MailPostDev& P, double* const (&partials)[K], double* scratch /* >= 5 K doubles */) {
    __shared__ int isLast;
    const int tid = threadIdx.x + threadIdx.y * blockDim.x;
    if (tid == 0) {
        // release the partials (agent scope: the other workgroups may sit on other XCDs with their own L2), take a ticket, acquire what the earlier tickets released
        const unsigned t = __hip_atomic_fetch_add(P.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        isLast = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!isLast) return;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                   // every thread of the last workgroup, not only the ticket holder
    const int lane = tid & (kWave - 1), wave = tid >> 6, n = gridDim.x;
    double t[K];
    if (tid < 256) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            t[k] = 0;
            for (int i = tid; i < n; i += 256) t[k] += __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)(partials[k] + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { t[k] = waveReduceSum(t[k]); if (lane == 0) scratch[k * 4 + wave] = t[k]; }
    }
    __syncthreads();
    if (tid < K) scratch[4 * K + tid] = ((scratch[tid * 4] + scratch[tid * 4 + 1]) + scratch[tid * 4 + 2]) + scratch[tid * 4 + 3];
    __syncthreads();
    const int nw = 2 * K;
    if (tid < P.world * nw) {
        const int r = tid / nw, w = tid % nw;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(scratch[4 * K + (w >> 1)]);
        const unsigned half = (w & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(P.dst[r] + w, ((unsigned long long)P.tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid == 0) __hip_atomic_store(P.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch (ordered by the kernel boundary)
}

// ---- per-kernel hipEvent timing (reference util.t:404-511) ---------------------------------------------
struct KernelTimer {
    bool enabled = false;
    // coarse: consecutive launches under the same name share ONE event pair -- the start of the first, the end of the last (recorded when another name begins or the
    // table is evaluated: in stream order that is right behind the last launch).  A loop of 400 PCGIteration launches then costs two event records instead of 800,
    // and its time is the loop's own (bench.py's roofline leg: the events of the per-launch mode cost ~1 % of a step).  OptAmd_PlanSetTiming(plan, 2).
    bool coarse = false;
    struct Rec { std::string name; hipEvent_t a, b; long count = 1; bool open = false; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    std::map<std::string, std::pair<long, double>> totals;   // name -> (count, ms), filled by evaluate()
    std::vector<std::string> order;

    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e;
    }
    void closeRun(hipStream_t s) { if (!recs.empty() && recs.back().open) { HIP_CHECK(hipEventRecord(recs.back().b, s)); recs.back().open = false; } }
    void begin(const char* name, hipStream_t s) {
        if (!enabled) return;
        if (coarse) {
            if (!recs.empty() && recs.back().open && recs.back().name == name) { ++recs.back().count; return; }
            closeRun(s);
        }
        Rec r{name, get(), get()};
        r.open = coarse;
        HIP_CHECK(hipEventRecord(r.a, s));
        recs.push_back(r);
        lastStream = s;
    }
    void end(hipStream_t s) {
        if (!enabled || coarse) return;
        HIP_CHECK(hipEventRecord(recs.back().b, s));
    }
    hipStream_t lastStream = nullptr;
    void reset() {
        for (auto& r : recs) { pool.push_back(r.a); pool.push_back(r.b); }
        recs.clear(); totals.clear(); order.clear();
    }
    void evaluate() {   // synchronises every pending event pair and folds it into `totals`
        closeRun(lastStream);
        for (auto& r : recs) {
            HIP_CHECK(hipEventSynchronize(r.b));
            float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
            auto it = totals.find(r.name);
            if (it == totals.end()) { totals[r.name] = {r.count, (double)ms}; order.push_back(r.name); }
            else { it->second.first += r.count; it->second.second += ms; }
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
    }
    // Same layout as the reference's Timer:evaluate (util.t:469-508): the table, the TIMING line (totals of the
    // PCGInit1* / PCGStep1* / PCGIteration / overall rows) and the per-iteration line that harness scripts grep.
    // "nonlinear" = time of the once-per-outer-iteration kernels per outer iteration, "linear" = time of the
    // PCG-loop kernels per PCG iteration (the reference derives both by matching launch counts, which amounts to the same).
    void print() const {
        printf("--------------------------------------------------------\n");
        printf("        Kernel        |   Count  |   Total   | Average \n");
        printf("----------------------+----------+-----------+----------\n");
        long nonLin = 0, lin = 0;
        for (auto& n : order) {
            auto& t = totals.at(n);
            printf("----------------------+----------+-----------+----------\n");
            printf(" %-20s |   %4ld   | %8.3fms| %7.4fms\n", n.c_str(), t.first, t.second, t.second / t.first);
            if (n.rfind("PCGInit1", 0) == 0 && n.find("_") == std::string::npos) nonLin = t.first;
            if (n == "PCGStep2" || n == "PCGIteration") lin = std::max(lin, t.first);
        }
        printf("--------------------------------------------------------\n");
        printf("TIMING ");
        for (auto& n : order)
            if (n.rfind("PCGInit1", 0) == 0 || n.rfind("PCGStep1", 0) == 0 || n.rfind("PCGStep3+PCGStep1", 0) == 0 || n == "PCGIteration" || n == "overall")
                printf("%f ", totals.at(n).second);
        printf("\n");
        double nonLinTotal = 0, linTotal = 0;
        for (auto& n : order) {
            auto& t = totals.at(n);
            if (n == "overall") continue;
            if (lin > 0 && t.first * 2 >= lin && t.first > nonLin * 2) linTotal += t.second; else nonLinTotal += t.second;
        }
        printf("Per-iter times ms (nonlinear,linear): %7.4f\t%7.4f\n", nonLin ? nonLinTotal / nonLin : 0.0, lin ? linTotal / lin : 0.0);
    }
    ~KernelTimer() { for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } for (auto e : pool) (void)hipEventDestroy(e); }
};

struct LaunchCtx {
    hipStream_t stream = nullptr;
    KernelTimer* timer = nullptr;
};
struct ScopedKernel {   // brackets one (logical) kernel launch with timing events + error check
    LaunchCtx& c;
    ScopedKernel(LaunchCtx& ctx, const char* name) : c(ctx) { if (c.timer) c.timer->begin(name, c.stream); }
    ~ScopedKernel() { HIP_CHECK(hipGetLastError()); if (c.timer) c.timer->end(c.stream); }
};

inline int divUp(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ void sincosT(float a, float* s, float* c) { sincosf(a, s, c); }
__device__ __forceinline__ void sincosT(double a, double* s, double* c) { sincos(a, s, c); }

}  // namespace optamd
