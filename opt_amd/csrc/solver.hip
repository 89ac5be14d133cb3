// Gauss-Newton / Levenberg-Marquardt outer loop + matrix-free PCG inner loop, generic over the energy.
//
// What it reproduces (reference API/src/solverGPUGaussNewton.t): init :956-1007, step :1016-1177 with the
// same kernel order, the same guards (alpha = 0 unless denominator > 0 :456-459, beta :544-547), the CERES
// guarded inverse :323-332, the LM diagonal / trust-region logic :631-664, 1119-1157, the residual reset
// every `residual_reset_period` iterations :1077-1086 and the q-based early out :1093-1102.
// What it changes (MI355X-first):
//  * the streaming kernels (Step2, Step3, LinearUpdate, ...) run over the FLAT unknown vector with 16-byte
//    accesses -- they never touch the exclude mask (see energy.h contract);
//  * global sums are per-workgroup partials in double, summed in index order by each consumer workgroup:
//    no same-address atomics, no hipMemset/hipMemcpy between kernels, bitwise reproducible;
//  * alphaNumerator <- betaNumerator (a D2D memcpy per iteration in the reference, :1091) is a two-slot
//    rotation written by workgroup 0 of Step3.
#include "solver.h"
#include <chrono>
#include <mutex>
#include <cmath>
#include <cstddef>
#include <cstring>

// a polite spin: pause on x86, yield on arm64, nothing elsewhere
static inline void cpuRelax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#endif
}

namespace optamd {

// ------------------------------------------------------------------------------------------------------
// 16-byte packs of opt_float
template <class T> struct Pack;
template <> struct alignas(16) Pack<float> { float v[4]; };
template <> struct alignas(16) Pack<double> { double v[2]; };
template <class T> struct PackN;
template <> struct PackN<float> { static constexpr int N = 4; typedef float vec __attribute__((ext_vector_type(4))); };
template <> struct PackN<double> { static constexpr int N = 2; typedef double vec __attribute__((ext_vector_type(2))); };

// Streaming (non-temporal) 16-byte accesses: every PCG vector is far larger than L2 + Infinity Cache and is
// touched once per kernel, so the hot streaming kernels ask the memory system not to retain the lines.
// Measured on MI355X (tools/microbench_stream.hip, PCGStep2 shape at 4096^2): plain 302 us, nt + 2 packs in
// flight per lane 270 us.
#ifndef S2_NT_LOAD
#define S2_NT_LOAD 1
#endif
template <class T> __device__ __forceinline__ Pack<T> ldnt(const T* base, long i) {
    typedef typename PackN<T>::vec V;
    const V v = S2_NT_LOAD ? __builtin_nontemporal_load((const V*)base + i) : ((const V*)base)[i];
    Pack<T> p;
#pragma unroll
    for (int k = 0; k < PackN<T>::N; ++k) p.v[k] = v[k];
    return p;
}
template <class T> __device__ __forceinline__ void stnt(T* base, long i, const Pack<T>& p) {
    typedef typename PackN<T>::vec V;
    V v;
#pragma unroll
    for (int k = 0; k < PackN<T>::N; ++k) v[k] = p.v[k];
    __builtin_nontemporal_store(v, (V*)base + i);
}

template <class T> __device__ __forceinline__ T guardedInvert(T x) {   // solver.t:323-332 (CERES)
    T s = T(1) + sqrt(x);
    return T(1) / (s * s);
}

// PCGInit1 (non-graph tail) / PCGInit1_Finish (graph): solver.t:384-392, 399-419.  r already holds -J^T F.
template <class T>
__global__ __launch_bounds__(kBlock) void k_initFinish(const T* __restrict__ r, const T* __restrict__ diag, T* __restrict__ pre,
                                                       T* __restrict__ p, T* __restrict__ delta, long nPacks, int usePre, int graphMode,
                                                       double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    constexpr int N = PackN<T>::N;
    double acc = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nPacks; i += (long)gridDim.x * blockDim.x) {
        Pack<T> R = ((const Pack<T>*)r)[i], D = ((const Pack<T>*)diag)[i], PR, PP, Z;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            T d = usePre ? D.v[k] : T(1);
            T pr = guardedInvert(d);
            if (graphMode && !usePre) pr = T(1);
            T pp = pr * R.v[k];
            PR.v[k] = pr; PP.v[k] = pp; Z.v[k] = T(0);
            acc += (double)(R.v[k] * pp);
        }
        ((Pack<T>*)pre)[i] = PR; ((Pack<T>*)p)[i] = PP; ((Pack<T>*)delta)[i] = Z;
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// total[0] = sum(partials[0..n))   (one workgroup)
__global__ __launch_bounds__(kBlock) void k_finalizeSum(const double* __restrict__ partials, int n, double* __restrict__ total) {
    __shared__ double scratch[kBlock / kWave + 1];
    double s = sumPartials(partials, n, scratch);
    if (threadIdx.x == 0) total[0] = s;
}

// total[k] = sum(partials_k[0..n_k)) for four reductions at once: workgroup k handles array k (slab mode: one launch
// before the single 4-double all-reduce of the fused PCG iteration)
struct Partials4 { const double* p[4]; int n[4]; };
__global__ __launch_bounds__(kBlock) void k_finalizeSum4(Partials4 in, double* __restrict__ total) {
    __shared__ double scratch[kBlock / kWave + 1];
    double s = sumPartials(in.p[blockIdx.x], in.n[blockIdx.x], scratch);
    if (threadIdx.x == 0) total[blockIdx.x] = s;
}

// PCGStep2: solver.t:446-489
template <class T, bool LM>
__global__ __launch_bounds__(kBlock) void k_step2(T* __restrict__ delta, const T* __restrict__ p, T* __restrict__ r, const T* __restrict__ Ap,
                                                  const T* __restrict__ pre /*nullptr -> 1*/, const T* __restrict__ b, T* __restrict__ z, long nPacks,
                                                  const double* __restrict__ aNumTotal, const double* __restrict__ aDenPartials, int nDen,
                                                  double* __restrict__ bNumPartials, double* __restrict__ qPartials) {
    __shared__ double scratch[kBlock / kWave + 1];
    constexpr int N = PackN<T>::N;
    const T aDen = (T)sumPartials(aDenPartials, nDen, scratch);
    const T aNum = (T)aNumTotal[0];
    const T alpha = (aDen > T(0)) ? aNum / aDen : T(0);   // guardDivisionByZero, solver.t:456-459
    double accB = 0, accQ = 0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x; i0 < nPacks; i0 += 2 * stride) {
        // two packs per lane in flight: issue all loads of both before the first use
        Pack<T> D[2], P[2], R[2], A[2], M[2], B[2], Z;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long i = i0 + u * stride;
            if (i < nPacks) {
                D[u] = ldnt(delta, i); P[u] = ldnt(p, i); R[u] = ldnt(r, i); A[u] = ldnt(Ap, i);
                if (pre) M[u] = ldnt(pre, i);
                if (LM) B[u] = ldnt(b, i);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long i = i0 + u * stride;
            if (i < nPacks) {
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    T dl = D[u].v[k] + alpha * P[u].v[k];
                    T rr = R[u].v[k] - alpha * A[u].v[k];
                    T m = pre ? M[u].v[k] : T(1);
                    T zz = m * rr;
                    D[u].v[k] = dl; R[u].v[k] = rr; Z.v[k] = zz;
                    accB += (double)(zz * rr);
                    if (LM) accQ += (double)(T(0.5) * (dl * (rr + B[u].v[k])));
                }
                stnt(delta, i, D[u]); stnt(r, i, R[u]); stnt(z, i, Z);      // non-temporal: nothing of this pass is read again before the next one has streamed past it
            }
        }
    }
    double t = blockReduceSum(accB, scratch);
    if (threadIdx.x == 0) bNumPartials[blockIdx.x] = t;
    if (LM) {
        double tq = blockReduceSum(accQ, scratch);
        if (threadIdx.x == 0) qPartials[blockIdx.x] = tq;
    }
}

// PCGStep2_1stHalf: solver.t:491-503
// deltaOut may be delta (in place) or another vector (the caller then decides later which of the two it keeps).  alphaNumerator is either a
// finished total (aNumTotal, nNum == 0) or partial sums this kernel adds up itself -- the same sumPartials over kBlock threads as
// k_finalizeSum, so the same bits, one launch less.
// pOwed / alphaOwed (may be null): a term alphaOwed[0] * pOwed that an earlier launch left owed to delta (EnergyOps::iterOwedTerm) is added first -- the reference's
// order of additions, one pass instead of two.
template <class T>
__global__ __launch_bounds__(kBlock) void k_step2FirstHalf(const T* delta, T* deltaOut, const T* __restrict__ p, long nPacks, const double* __restrict__ aNumTotal,
                                                           const double* __restrict__ aNumPartials, int nNum, const double* __restrict__ aDenPartials, int nDen,
                                                           const T* __restrict__ pOwed = nullptr, const T* __restrict__ alphaOwed = nullptr) {
    __shared__ double scratch[kBlock / kWave + 1];
    constexpr int N = PackN<T>::N;
    const T aDen = (T)sumPartials(aDenPartials, nDen, scratch);
    const T aNum = nNum > 0 ? (T)sumPartials(aNumPartials, nNum, scratch) : (T)aNumTotal[0];
    const T alpha = (aDen > T(0)) ? aNum / aDen : T(0);
    const T a2 = pOwed ? alphaOwed[0] : T(0);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nPacks; i += (long)gridDim.x * blockDim.x) {
        Pack<T> D = ((const Pack<T>*)delta)[i], P = ((const Pack<T>*)p)[i];
        if (pOwed) {
            const Pack<T> P2 = ((const Pack<T>*)pOwed)[i];
#pragma unroll
            for (int k = 0; k < N; ++k) D.v[k] = D.v[k] + a2 * P2.v[k];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) D.v[k] = D.v[k] + alpha * P.v[k];
        ((Pack<T>*)deltaOut)[i] = D;
    }
}

// Completion stamp for PcgSolver::drain(): everything enqueued before it has finished when the host sees the value.
__global__ void k_stamp(unsigned long long* flag, unsigned long long v) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// PCGStep2_2ndHalf: solver.t:505-534
template <class T>
__global__ __launch_bounds__(kBlock) void k_step2SecondHalf(const T* __restrict__ delta, T* __restrict__ r, const T* __restrict__ Adelta, const T* __restrict__ b,
                                                            const T* __restrict__ pre, T* __restrict__ z, long nPacks, double* __restrict__ bNumPartials,
                                                            double* __restrict__ qPartials) {
    __shared__ double scratch[kBlock / kWave + 1];
    constexpr int N = PackN<T>::N;
    double accB = 0, accQ = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nPacks; i += (long)gridDim.x * blockDim.x) {
        Pack<T> D = ((const Pack<T>*)delta)[i], A = ((const Pack<T>*)Adelta)[i], B = ((const Pack<T>*)b)[i], M, R, Z;
        if (pre) M = ((const Pack<T>*)pre)[i];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            T rr = B.v[k] - A.v[k];
            T m = pre ? M.v[k] : T(1);
            T zz = m * rr;
            R.v[k] = rr; Z.v[k] = zz;
            accB += (double)(zz * rr);
            accQ += (double)(T(0.5) * (D.v[k] * (rr + B.v[k])));
        }
        ((Pack<T>*)r)[i] = R; ((Pack<T>*)z)[i] = Z;
    }
    double t = blockReduceSum(accB, scratch);
    if (threadIdx.x == 0) bNumPartials[blockIdx.x] = t;
    double tq = blockReduceSum(accQ, scratch);
    if (threadIdx.x == 0) qPartials[blockIdx.x] = tq;
}

// PCGStep3: solver.t:537-550; workgroup 0 also publishes betaNumerator as the next alphaNumerator (:1091)
template <class T>
__global__ __launch_bounds__(kBlock) void k_step3(const T* __restrict__ z, T* __restrict__ p, long nPacks, const double* __restrict__ bNumPartials, int nB,
                                                  const double* __restrict__ aNumOld, double* __restrict__ aNumNext) {
    __shared__ double scratch[kBlock / kWave + 1];
    constexpr int N = PackN<T>::N;
    const double bSum = sumPartials(bNumPartials, nB, scratch);
    const T rDotzNew = (T)bSum, rDotzOld = (T)aNumOld[0];
    const T beta = (rDotzOld > T(0)) ? rDotzNew / rDotzOld : T(0);
    if (blockIdx.x == 0 && threadIdx.x == 0) aNumNext[0] = bSum;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nPacks; i += (long)gridDim.x * blockDim.x) {
        Pack<T> Z = ((const Pack<T>*)z)[i], P = ((Pack<T>*)p)[i];
#pragma unroll
        for (int k = 0; k < N; ++k) P.v[k] = Z.v[k] + beta * P.v[k];
        ((Pack<T>*)p)[i] = P;
    }
}

// PCGLinearUpdate / revertUpdate / savePreviousUnknowns on one unknown image (caller-owned array, not padded)
template <class T>
__global__ __launch_bounds__(kBlock) void k_axpyImage(T* __restrict__ X, const T* __restrict__ d, long n) {   // X += d   (solver.t:552-557)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) X[i] = X[i] + d[i];
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_saveAndUpdate(T* __restrict__ X, T* __restrict__ prev, const T* __restrict__ d, long n) {   // prev = X; X += d  (LM: solver.t:1113-1114 in one pass)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const T x = X[i]; prev[i] = x; X[i] = x + d[i]; }
}
template <class T>
__global__ __launch_bounds__(kBlock) void k_copy(T* __restrict__ dst, const T* __restrict__ src, long n) {    // solver.t:559-564, 573-578, 624-629
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

// PCGComputeCtC + PCGFinalizeDiagonal: solver.t:616-622, 631-664.  On entry CtC holds raw diag(J^T J).
// INIT: the kernel also does what PCGInit1's tail does in an LM step before it (k_initFinish: delta = 0, the guarded-inverse preconditioner -- needed only to
// seed SSq at the first outer iteration, solver.t:635 -- and a p that this kernel overwrites anyway): one pass over the vectors instead of two or three.
template <class T, bool INIT>
__global__ __launch_bounds__(kBlock) void k_finalizeDiagonal(T* __restrict__ CtC, T* __restrict__ SSq, const T* __restrict__ r, T* __restrict__ delta,
                                                             T* __restrict__ pre, T* __restrict__ b, T* __restrict__ p, long nPacks, T radius, T minLm, T maxLm,
                                                             double* __restrict__ dPartials, double* __restrict__ qPartials, int usePre, int graphMode, int saveSSq) {
    __shared__ double scratch[kBlock / kWave + 1];
    constexpr int N = PackN<T>::N;
    const T invRadius = T(1) / radius;
    double accD = 0, accQ = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nPacks; i += (long)gridDim.x * blockDim.x) {
        Pack<T> C = ((Pack<T>*)CtC)[i], R = ((const Pack<T>*)r)[i], S, D, M, P;
        if (INIT) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                T pr = guardedInvert(usePre ? C.v[k] : T(1));      // k_initFinish
                if (graphMode && !usePre) pr = T(1);
                S.v[k] = pr; D.v[k] = T(0);
            }
            if (saveSSq) ((Pack<T>*)SSq)[i] = S;                   // PCGSaveSSq (first outer iteration)
            else S = ((const Pack<T>*)SSq)[i];
            ((Pack<T>*)delta)[i] = D;
        } else { S = ((const Pack<T>*)SSq)[i]; D = ((const Pack<T>*)delta)[i]; }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            T unclamped = C.v[k] * invRadius;                 // computeCtC: diag(J^T J) / radius (o.t:2277-2279)
            T invS = T(1) / S.v[k];
            T clampMul = invS / radius;
            T lo = minLm * clampMul, hi = maxLm * clampMul;
            T c = fmin(fmax(unclamped, lo), hi);
            C.v[k] = c;
            T m = T(1) / (c + radius * unclamped);
            M.v[k] = m;
            T pp = m * R.v[k];
            P.v[k] = pp;
            accD += (double)(R.v[k] * pp);
            accQ += (double)(T(0.5) * (D.v[k] * (R.v[k] + R.v[k])));
        }
        ((Pack<T>*)CtC)[i] = C; ((Pack<T>*)pre)[i] = M; ((Pack<T>*)b)[i] = R; ((Pack<T>*)p)[i] = P;
    }
    double t = blockReduceSum(accD, scratch);
    if (threadIdx.x == 0) dPartials[blockIdx.x] = t;
    double tq = blockReduceSum(accQ, scratch);
    if (threadIdx.x == 0) qPartials[blockIdx.x] = tq;
}

// ------------------------------------------------------------------------------------------------------
template <class T>
struct PcgSolver : SolverBase {
    std::unique_ptr<EnergyOps<T>> E;
    bool lm;
    bool patch = false; int patchSweep = 0;   // kind "patchGaussNewtonGPU": block-local sweeps instead of the global PCG loop (stepPatch)
    hipStream_t stream = nullptr;
    LaunchCtx ctx;
    long n = 0, nPad = 0, nPacks = 0;
    int streamGrid = 0;
    // PlanData vectors (solver.t:173-185); LM-only ones are allocated for LM plans only; `g` is never used by the reference
    T *delta = nullptr, *r = nullptr, *b = nullptr, *Adelta = nullptr, *z = nullptr, *p = nullptr, *Ap_X = nullptr, *CtC = nullptr, *preconditioner = nullptr,
      *SSq = nullptr, *prevX = nullptr;
    T* p2 = nullptr;                    // second search-direction buffer for the fused PCGStep3+PCGStep1 kernel
    T *r2 = nullptr, *Ap2 = nullptr;    // second r / Ap buffers for the single-kernel PCG iteration (z doubles as nothing there)
    T* delta2 = nullptr;                // second delta buffer of the LM single-kernel loop (allocated on first use)
    bool oneKernel = true;              // OPT_AMD_ONEKERNEL=0: use the Step1(+3)/Step2 pair instead of one kernel per PCG iteration
    bool oneKernelLM = true;            // OPT_AMD_ONEKERNEL_LM=0: the same switch for the Levenberg-Marquardt loop only
    Reduction setS[2][4];               // ping-pong {alphaNum, alphaDen, s2, s3} of the single-kernel iteration
    bool unknownsUpdated = false;       // this step's PCGLinearUpdate was folded into the end of the PCG loop (EnergyOps::finishUpdate)
    bool keepReferenceP = false;        // run the (dead) last PCGStep3 so that `p` matches the reference after a step
    std::vector<void*> allocs;
    Reduction redA, redB, redQ, redC;   // alpha denominator, beta numerator, q, cost / init numerator
    Reduction redQ2;                    // second Q buffer: the LM single-kernel loop enqueues launch k + 1 (which writes Q_k) before the host has read Q_{k-1}
    Reduction redQR;                    // pinned: Q of the split residual reset (its own buffer: the reset is enqueued while the host may still poll redQ / redQ2)
    unsigned long long* stampFlag = nullptr; unsigned long long stampSeq = 0; bool pollSync = true;   // drain(): false (only after a stamp failed to arrive) -> hipStreamSynchronize
    Reduction redMH, redCH;             // pinned: model cost / cost partials that only the host sums (no copy kernel between the producer and the read)
    unsigned launchTag = 0;             // tags of the Q partials the single-kernel LM launches deliver as self-validating words (common.h storeTaggedPartial)
    bool taggedQ = true;                // Q of the single-kernel LM launches as tagged words the host polls; false (only after a tag failed to arrive): plain partials + an event in the stream
    double* scal = nullptr;             // device: [0],[1] alphaNumerator ping-pong, [2..5] slab totals
    double* scal4[2] = {nullptr, nullptr};   // device: all-reduced {alphaNum, alphaDen, s2, s3} of the single-kernel iteration (slab mode), ping-pong
    int aSlot = 0;
    double* hostBuf = nullptr;          // pinned
    double* hostBufQ = nullptr; hipEvent_t qEvent = nullptr; int qCount = 0; const double* qSrc = nullptr;   // pinned buffer + event of the overlapped q fetch
    T prevCost = 0;
    T trust_region_radius = 0, radius_decrease_factor = 0, min_lm_diagonal = 0, max_lm_diagonal = 0;   // pd.parameters (o.t:933-938)
    hipEvent_t overallStart = nullptr; bool overallOpen = false;
    OptAmd_SlabComm comm{};
    OptAmd_SlabCommExt commExt{};       // the communicator's optional fast paths (OptAmd_PlanSetSlabExt); all null unless set
    bool distributed = false;

    T* allocVec() {
        T* v; HIP_CHECK(hipMalloc((void**)&v, nPad * sizeof(T))); HIP_CHECK(hipMemset(v, 0, nPad * sizeof(T)));   // zero-initialised like o.t:627-632
        allocs.push_back(v); return v;
    }
    Reduction allocRed() {
        Reduction R; HIP_CHECK(hipMalloc((void**)&R.partials, kMaxPartials * sizeof(double))); HIP_CHECK(hipMemset(R.partials, 0, kMaxPartials * sizeof(double)));
        allocs.push_back(R.partials); return R;
    }
    PcgSolver(EnergyOps<T>* e, bool useLM, bool timing, int verb) : E(e), lm(useLM) {
        verbosity = verb; timer.enabled = timing;
        HIP_CHECK(hipStreamCreate(&stream));   // blocking stream: ordered against the caller's null-stream work
        ctx.stream = stream; ctx.timer = timing ? &timer : nullptr;
        n = E->nScalars; nPad = (n + 3) / 4 * 4; nPacks = nPad / PackN<T>::N;
        int dev = 0, cus = 256; HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        long want = (nPacks + kBlock - 1) / kBlock;
        streamGrid = (int)std::max<long>(1, std::min<long>(want, std::min<long>(kMaxPartials, (long)cus * 8)));
        delta = allocVec(); r = allocVec(); z = allocVec(); p = allocVec(); Ap_X = allocVec(); CtC = allocVec(); preconditioner = allocVec();
        if (lm) { b = allocVec(); Adelta = allocVec(); SSq = allocVec(); prevX = allocVec(); }
        p2 = allocVec();
        if (const char* e = getenv("OPT_AMD_ONEKERNEL")) oneKernel = atoi(e) != 0;
        if (const char* e = getenv("OPT_AMD_DELTA_TRIAL")) trialMode = atoi(e);
        if (const char* e = getenv("OPT_AMD_ONEKERNEL_LM")) oneKernelLM = atoi(e) != 0;
        r2 = allocVec(); Ap2 = allocVec();                // second r / A p buffers of the single-kernel iterations (kernels that keep A p in memory read the old one on a halo)
        for (auto& st : setS) for (auto& R : st) R = allocRed();
        redA = allocRed(); redB = allocRed(); redC = allocRed();
        // Q (solver.t:483-485, 1093-1102) is read by the host once per LM iteration and by no kernel: its partials go straight to pinned host memory
        // (<= 16 KB of posted writes per launch) instead of through a device buffer and a copy kernel per iteration (830 copyBuffer launches, 7 % of config 3's GPU time)
        // With the single-kernel loop each partial is two tagged words (2 x kMaxPartials slots), which the host polls: no event packet in the stream either.
        for (Reduction* R : {&redQ, &redQ2, &redQR, &redMH, &redCH}) { HIP_CHECK(hipHostMalloc((void**)&R->partials, 2 * kMaxPartials * sizeof(double))); memset(R->partials, 0, 2 * kMaxPartials * sizeof(double)); R->hostVisible = true; }
        HIP_CHECK(hipHostMalloc((void**)&stampFlag, 64)); *stampFlag = 0;
        HIP_CHECK(hipMalloc((void**)&scal, 16 * sizeof(double))); HIP_CHECK(hipMemset(scal, 0, 16 * sizeof(double))); allocs.push_back(scal);
        scal4[0] = scal + 8; scal4[1] = scal + 12;
        HIP_CHECK(hipHostMalloc((void**)&hostBuf, 2 * kMaxPartials * sizeof(double)));      // room for two reductions read in one go (LM: model cost + new cost)
        E->slab = Slab{};
    }
    ~PcgSolver() override {
        (void)hipStreamSynchronize(stream);
        dropLease();
        for (void* a : allocs) (void)hipFree(a);
        if (hostBuf) (void)hipHostFree(hostBuf);
        if (redQ.partials) (void)hipHostFree(redQ.partials);
        if (redQ2.partials) (void)hipHostFree(redQ2.partials);
        if (redMH.partials) (void)hipHostFree(redMH.partials);
        if (redQR.partials) (void)hipHostFree(redQR.partials);
        if (stampFlag) (void)hipHostFree(stampFlag);
        if (lmBreak) (void)hipHostFree(lmBreak);
        if (onChipTrace) (void)hipFree(onChipTrace);
        if (redCH.partials) (void)hipHostFree(redCH.partials);
        for (Reduction& R : costRing) if (R.partials) (void)hipHostFree(R.partials);
        for (hipEvent_t e : trialEv) if (e) (void)hipEventDestroy(e);
        if (hostBufQ) { (void)hipHostFree(hostBufQ); (void)hipEventDestroy(qEvent); }
        (void)hipStreamDestroy(stream);
    }

    // ---- co-residency of the persistent kernels (EnergyOps::pcgSolveOnChip): their workgroups wait for each other, so the whole grid has to be resident ----
    // (i)  Two plans of ONE process stepped from two host threads would interleave two such grids on the CUs: a process-wide lease per device serialises the
    //      on-chip launches -- taken before the launch, dropped when this step has drained the stream; a plan that cannot get it within 50 ms runs this step
    //      on the streaming kernels.
    // (ii) A foreign tenant (another process, another library's long kernel) is caught by the kernel itself: the waits of its FIRST phase are bounded by
    //      10 ms (firstTicks) -- every workgroup posts its words before it waits, so passing that wait proves the grid resident; nothing has been written by then.
    // (iii) After a time-out the step is redone by the streaming kernels and the plan stays on them for `onChipBackoff` clean steps (8, then 16, 32 ... 1024), then
    //      tries the chip again (EnergyOps::onChipRearm): a tenant that has left does not cost the plan its fast path for life.
    static std::timed_mutex& chipLease() { static std::timed_mutex m[64]; int dev = 0; (void)hipGetDevice(&dev); return m[(unsigned)dev % 64u]; }
    bool leaseHeld = false;
    bool takeLease() { if (leaseHeld) return true; leaseHeld = chipLease().try_lock_for(std::chrono::milliseconds(50)); return leaseHeld; }
    void dropLease() { if (leaseHeld) { chipLease().unlock(); leaseHeld = false; } }
    int onChipFailures = 0, onChipBackoff = 0, onChipCleanSteps = 0;
    bool boundForSolve = false;      // bind() has run inside the current Opt_ProblemSolve (SolverBase::insideSolve)
    bool jtfReady = false;           // the pass that computed the last step's cost also ran this step's PCGInit1 (EnergyOps::evalCostAndJTFInit; only inside Opt_ProblemSolve)
    // Deferred Gauss-Newton steps (inside Opt_ProblemSolve, kernel sets with evalCostAndJTFInit): nothing a step computes steers the next one -- the cost is only reported --
    // so up to kDefer - 1 steps are enqueued back to back and their costs (one pinned partials buffer each) and on-chip verdicts (one word each) are read at the next drain.
    // Where delta lives (round 6).  delta is the one vector the Gauss-Newton loop reads AND writes, and the time of a launch follows the region the allocator put it in
    // (profiles/NOTES.md: 173 / 176 / 191 us at 4096^2 for the same binary in one process, region by region; the ring, the flag bytes and the caller's arrays do not matter).
    // The first long linear solve of a large single-GPU plan therefore tries kTrialCands allocations of it: after launch kTrialFirst delta is copied into a fresh vector
    // every kTrialWindow launches (a copy: no bit changes), each window is timed with one event pair, the loop goes on in the fastest and the others are freed.
    // OPT_AMD_DELTA_TRIAL=0: off, 2: report on stderr.
    static constexpr int kTrialCands = 4, kTrialWindow = 6, kTrialFirst = 4;
    int trialMode = 1, trialPhase = 0, trialAttempts = 0; std::vector<T*> trialVecs; std::vector<float> trialMs; hipEvent_t trialEv[2] = {nullptr, nullptr};
    void trialDrop(T* keep) {      // free every trial vector but `keep`
        for (T* v : trialVecs) if (v != keep) { for (auto it = allocs.begin(); it != allocs.end(); ++it) if (*it == (void*)v) { allocs.erase(it); break; } (void)hipFree(v); }
        trialVecs.clear(); trialMs.clear();
    }
    void deltaTrial(int lIter) {
        if (lIter == 0) {
            if (trialPhase == 1) { trialDrop(delta); trialPhase = 0; }      // the last solve ended inside the trial: it stayed where it had got to
            if (trialPhase == 0 && trialAttempts < 2 && sp.lIterations > kTrialFirst + kTrialCands * kTrialWindow) { trialPhase = 1; ++trialAttempts; trialVecs.assign(1, delta); trialMs.clear(); }
        }
        if (trialPhase != 1 || lIter < kTrialFirst || (lIter - kTrialFirst) % kTrialWindow != 0) return;
        const int w = (lIter - kTrialFirst) / kTrialWindow;      // the window about to start
        if (!trialEv[0]) { HIP_CHECK(hipEventCreate(&trialEv[0])); HIP_CHECK(hipEventCreate(&trialEv[1])); }
        if (w > 0) {
            float ms = 0;
            HIP_CHECK(hipEventRecord(trialEv[1], stream)); HIP_CHECK(hipEventSynchronize(trialEv[1])); HIP_CHECK(hipEventElapsedTime(&ms, trialEv[0], trialEv[1]));
            trialMs.push_back(ms);
        }
        auto moveTo = [&](T* to) { if (to != delta) { HIP_CHECK(hipMemcpyAsync(to, delta, nPad * sizeof(T), hipMemcpyDeviceToDevice, stream)); delta = to; } };
        if (w == 0) { HIP_CHECK(hipEventRecord(trialEv[0], stream)); return; }
        if (w < kTrialCands) {
            T* v = nullptr;
            if (hipMalloc((void**)&v, nPad * sizeof(T)) == hipSuccess) {
                allocs.push_back(v); trialVecs.push_back(v);
                moveTo(v);
                HIP_CHECK(hipEventRecord(trialEv[0], stream));
                return;
            }
            (void)hipGetLastError();      // (no room for another: decide among those timed so far)
        }
        int best = 0;
        for (int i = 1; i < (int)trialMs.size(); ++i) if (trialMs[i] < trialMs[best]) best = i;
        if (trialMode > 1) { fprintf(stderr, "Opt(amd): delta placement trial, ms per %d launches:", kTrialWindow); for (float m : trialMs) fprintf(stderr, " %.3f", m); fprintf(stderr, " -> %d\n", best); }
        moveTo(trialVecs[(size_t)best]);
        HIP_CHECK(hipStreamSynchronize(stream));      // (the copy out of a vector about to be freed has finished)
        trialDrop(delta);
        trialPhase = 2;
    }
    static constexpr int kDefer = 8;
    struct PendingStep { int slot, step; bool onChip; };
    std::vector<PendingStep> pendingSteps;
    Reduction costRing[kDefer];
    bool onChipAllowed() const { return onChipOk && sp.amd_onchip != 0 && sp.amd_reference_order == 0; }
    bool singleKernelAllowed() const { return oneKernel && sp.amd_reference_order == 0; }
    double* lmBreak = nullptr;          // pinned: {iteration + 1, zeta} of an on-chip LM solve's q early-out (OnChipLm::breakInfo)
    bool onChipOk = true, usedOnChip = false, onChipFellBack = false, lastStepOnChip = false; double* onChipTrace = nullptr; int onChipTraceCap = 0;      // EnergyOps::pcgSolveOnChip
    // true iff `mine` holds on every rank: one all-reduce of a count and one read-back (once per Gauss-Newton step in slab mode)
    bool allRanksAgree(bool mine) {
        hostBuf[0] = mine ? 0.0 : 1.0;
        HIP_CHECK(hipMemcpyAsync(scal + 5, hostBuf, sizeof(double), hipMemcpyHostToDevice, stream));
        comm.allReduceSum(comm.ctx, scal + 5, 1, (void*)stream);
        HIP_CHECK(hipMemcpyAsync(hostBuf, scal + 5, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        return hostBuf[0] == 0.0;
    }
    // ---- reductions ---------------------------------------------------------------------------------
    // Host value of a reduction (blocking D2H like the reference's computeCost / fetchQ, solver.t:790-814)
    double hostSum(const Reduction& R) {
        if (distributed) {
            reduceAcross(&R, 1, scal + 2);
            HIP_CHECK(hipMemcpyAsync(hostBuf, scal + 2, sizeof(double), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            return hostBuf[0];
        }
        if (R.hostVisible) { drain(); double s = 0; for (int i = 0; i < R.n; ++i) s += R.partials[i]; return s; }
        HIP_CHECK(hipMemcpyAsync(hostBuf, R.partials, R.n * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        double s = 0; for (int i = 0; i < R.n; ++i) s += hostBuf[i];
        return s;
    }
    // The same value without draining the stream: begin enqueues the copy and an event, end waits for that event only.
    void beginHostSum(const Reduction& R) {
        if (!hostBufQ) { HIP_CHECK(hipHostMalloc((void**)&hostBufQ, kMaxPartials * sizeof(double))); HIP_CHECK(hipEventCreateWithFlags(&qEvent, hipEventDisableTiming)); }
        if (distributed) {
            reduceAcross(&R, 1, scal + 2);
            HIP_CHECK(hipMemcpyAsync(hostBufQ, scal + 2, sizeof(double), hipMemcpyDeviceToHost, stream));
            qCount = 1;
            qSrc = hostBufQ;
        } else if (R.hostVisible) {
            qSrc = R.partials; qCount = R.n;      // the producer wrote pinned host memory: the event is all that is needed
        } else {
            HIP_CHECK(hipMemcpyAsync(hostBufQ, R.partials, R.n * sizeof(double), hipMemcpyDeviceToHost, stream));
            qCount = R.n; qSrc = hostBufQ;
        }
        HIP_CHECK(hipEventRecord(qEvent, stream));
    }
    double endHostSum() {
        HIP_CHECK(hipEventSynchronize(qEvent));
        double s = 0; for (int i = 0; i < qCount; ++i) s += qSrc[i];
        return s;
    }
    // Everything enqueued so far has finished and what it wrote to pinned memory is readable.  A one-thread kernel stamps a pinned word and the host spins
    // on it: no barrier packet with a completion signal in the queue and no wake-up through the runtime (the latency-bound LM steps drain twice or more).
    void drain() {
        if (!pollSync) { HIP_CHECK(hipStreamSynchronize(stream)); return; }
        const unsigned long long v = ++stampSeq;
        k_stamp<<<1, kWave, 0, stream>>>(stampFlag, v);
        unsigned long spins = 0;
        while (__atomic_load_n(stampFlag, __ATOMIC_ACQUIRE) != v) {
            if ((++spins & 0x3fff) == 0) {
                const hipError_t e = hipStreamQuery(stream);
                if (e != hipSuccess && e != hipErrorNotReady) HIP_CHECK(e);
                // The stream reports idle but the stamp has not been seen: fall back to the runtime's own completion (which also makes the device's
                // writes to pinned memory visible) and look once more; only a stamp that is still missing then is an error -- and polling is given up
                // for this plan rather than the process.
                if (e == hipSuccess) {
                    HIP_CHECK(hipStreamSynchronize(stream));
                    if (__atomic_load_n(stampFlag, __ATOMIC_ACQUIRE) != v) {
                        fprintf(stderr, "Opt(amd): completion stamp %llu not visible after a stream synchronise; switching this plan to hipStreamSynchronize\n", v);
                        pollSync = false;
                    }
                    return;
                }
            }
            cpuRelax();
        }
    }
    // Sum of a host-visible reduction whose producer wrote tagged word pairs: spin on pinned memory until every pair carries `tag`; nothing was put in the
    // stream for it.  The stream is queried now and then so that a faulted or never-issued producer ends in an error message instead of a hang.
    double pollTaggedSum(const Reduction& R, unsigned tag) {
        const unsigned long long* w = reinterpret_cast<const unsigned long long*>(R.partials);
        double s = 0; unsigned long spins = 0; bool synced = false;
        for (int i = 0; i < R.n; ++i) {
            unsigned long long lo, hi;
            for (;;) {
                lo = __atomic_load_n(w + 2 * i, __ATOMIC_ACQUIRE); hi = __atomic_load_n(w + 2 * i + 1, __ATOMIC_ACQUIRE);
                if ((unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag) break;
                if ((++spins & 0x3fff) == 0) {
                    const hipError_t e = hipStreamQuery(stream);
                    if (e != hipSuccess && e != hipErrorNotReady) HIP_CHECK(e);
                    if (e == hipSuccess) {      // idle stream, word not seen: one real synchronise and one more look before calling it an error
                        if (!synced) { HIP_CHECK(hipStreamSynchronize(stream)); synced = true; continue; }
                        fprintf(stderr, "Opt(amd): Q partial %d of tag %u never arrived (the producing launch was not issued or faulted); this early-out test is skipped and the plan goes back to reading Q through the stream\n", i, tag);
                        return std::nan("");
                    }
                }
                cpuRelax();
            }
            const unsigned long long bits = (lo & 0xffffffffull) | (hi << 32);
            double v; memcpy(&v, &bits, sizeof v); s += v;
        }
        return s;
    }
    // dst[i] = sum over ranks of sum(Rs[i].partials): one launch if the communicator folds the local reduction in (allReducePartials)
    void reduceAcross(const Reduction* Rs, int cnt, double* dst) {
        if (commExt.allReducePartials) {
            const double* ps[8]; int ns[8];
            for (int i = 0; i < cnt; ++i) { ps[i] = Rs[i].partials; ns[i] = Rs[i].n; }
            commExt.allReducePartials(comm.ctx, ps, ns, cnt, dst, (void*)stream);
            return;
        }
        if (cnt == 4) {
            Partials4 in4; for (int i = 0; i < 4; ++i) { in4.p[i] = Rs[i].partials; in4.n[i] = Rs[i].n; }
            k_finalizeSum4<<<4, kBlock, 0, stream>>>(in4, dst);
        } else for (int i = 0; i < cnt; ++i) k_finalizeSum<<<1, kBlock, 0, stream>>>(Rs[i].partials, Rs[i].n, dst + i);
        comm.allReduceSum(comm.ctx, dst, cnt, (void*)stream);
    }
    // What device consumers should sum: the partials themselves, or (slab mode) the all-reduced total.
    Reduction forConsumers(const Reduction& R, int slot) {
        if (!distributed) return R;
        double* tot = scal + 3 + slot;
        reduceAcross(&R, 1, tot);
        Reduction out; out.partials = tot; out.n = 1; return out;
    }
    void finalizeTo(const Reduction& R, double* dst) {
        ScopedKernel k(ctx, "finalizeSum");
        if (distributed) reduceAcross(&R, 1, dst);
        else k_finalizeSum<<<1, kBlock, 0, stream>>>(R.partials, R.n, dst);
    }

    // ---- halo exchange (slab mode) ------------------------------------------------------------------
    void exchangeRows(const std::vector<T*>& bases) {   // one grouped exchange for all unknown images
        const Slab& s = E->slab;
        const void* su[8]; const void* sd[8]; void* ru[8]; void* rd[8]; long bytes[8];
        const int nb = (int)bases.size();
        for (int i = 0; i < nb; ++i) {
            const long rs = E->rowScalars(i % (int)E->unknowns.size()); T* base = bases[i];   // bases: one per unknown image, possibly for several vectors
            su[i] = base + (long)s.yBegin * rs; sd[i] = base + (long)(s.yEnd - s.ghost) * rs;          // my first / last `ghost` owned rows
            ru[i] = base + (long)(s.yBegin - s.ghost) * rs; rd[i] = base + (long)s.yEnd * rs; bytes[i] = (long)s.ghost * rs * (long)sizeof(T);
        }
        comm.haloExchange(comm.ctx, nb, su, sd, ru, rd, bytes, (void*)stream);
    }
    void exchangeVector(T* v) {
        if (!distributed) return;
        std::vector<T*> bases;
        for (size_t i = 0; i < E->unknowns.size(); ++i) bases.push_back(v + E->unknowns[i].offset);
        exchangeRows(bases);
    }
    void exchangeUnknowns() {
        if (!distributed) return;
        std::vector<T*> bases;
        for (size_t i = 0; i < E->unknowns.size(); ++i) bases.push_back(E->unknownPtr((int)i));
        exchangeRows(bases);
    }

    // ---- pieces ---------------------------------------------------------------------------------------
    T computeCost() {   // solver.t:790-797
        Reduction& R = distributed ? redC : redCH;   // single GPU: the partials land in pinned memory and the host sums them after one drain
        E->evalCost(R, ctx);
        return (T)hostSum(R);
    }
    void imageOp(int kind) {   // 0: X += delta, 1: prevX = X, 2: X = prevX
        for (size_t i = 0; i < E->unknowns.size(); ++i) {
            const auto& u = E->unknowns[i];
            long cnt = u.elems * u.channels;
            int grid = (int)std::max<long>(1, std::min<long>((cnt + kBlock - 1) / kBlock, 4096));
            T* X = E->unknownPtr((int)i);
            if (kind == 0) { ScopedKernel k(ctx, "PCGLinearUpdate"); k_axpyImage<T><<<grid, kBlock, 0, stream>>>(X, delta + u.offset, cnt); }
            else if (kind == 3) { ScopedKernel k(ctx, "PCGLinearUpdate"); k_saveAndUpdate<T><<<grid, kBlock, 0, stream>>>(X, prevX + u.offset, delta + u.offset, cnt); }   // 1 then 0
            else if (kind == 1) { ScopedKernel k(ctx, "savePreviousUnknowns"); k_copy<T><<<grid, kBlock, 0, stream>>>(prevX + u.offset, X, cnt); }
            else { ScopedKernel k(ctx, "revertUpdate"); k_copy<T><<<grid, kBlock, 0, stream>>>(X, prevX + u.offset, cnt); }
        }
    }
    void record(int lIter, const Reduction& aDenR, const Reduction& bNumR, double q) {
        if (!traceEnabled) return;
        double aDen = hostSum(aDenR), bNum = distributed ? 0.0 : hostSum(bNumR), aNum = 0;
        if (distributed) { HIP_CHECK(hipMemcpyAsync(hostBuf, bNumR.partials, sizeof(double), hipMemcpyDeviceToHost, stream)); HIP_CHECK(hipStreamSynchronize(stream)); bNum = hostBuf[0]; }
        HIP_CHECK(hipMemcpyAsync(hostBuf, scal + aSlot, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        aNum = hostBuf[0];
        trace.insert(trace.end(), {(double)sp.nIter, (double)lIter, aNum, aDen, bNum, q});
    }

    // ---- PCG loop as one kernel per iteration (energy.h PcgIterArgs); returns false if the energy has no such kernel ----
    bool runSingleKernelLoop(const T* preArg) {
        // Row slabs: the on-chip solve runs on ALL ranks or on none (their kernels wait for each other): every rank says whether it could, the communicator adds it up.
        bool slabOnChip = false;
        if (distributed) {
            slabOnChip = allRanksAgree(onChipAllowed() && preArg && sp.lIterations > 0 && !traceEnabled && E->slabOnChipAvailable(sp.lIterations));
            if (!slabOnChip && !E->slabIterationAvailable()) return false;      // (before anything is exchanged: the three-kernel loop needs r = 0 on ghost rows)
            if (slabOnChip) {      // the kernel reads r_0, p_0 of its halo rows from the ghost rows
                (void)takeLease();      // (agreed collectively: this rank launches whether or not another plan of this process holds the chip)
                exchangeVector(r); exchangeVector(p);
                // A rank whose kernel set refuses after the vote (it should not: the vote covers the communicator's capacity and error state) launches nothing; its
                // peers' waits then time out, and the verdict below makes every rank redo the step with the streaming loop -- a library does not exit().
                const bool refused = !E->pcgSolveOnChip(r, p, delta, sp.lIterations, nullptr, nullptr, ctx);
                if (refused) fprintf(stderr, "Opt(amd): the slab on-chip solve was agreed on but refused by this rank's kernel set; the step will be redone by the streaming loop\n");
                // all ranks keep their update or none does: the ranks' verdicts (0 fine / 1 a wait timed out) are all-reduced on the device, PCGLinearUpdate checks the sum
                E->onChipVerdict(scal + 5, refused, ctx);
                comm.allReduceSum(comm.ctx, scal + 5, 1, (void*)stream);
                E->onChipApply(delta, scal + 5, refused, ctx);
                usedOnChip = true; unknownsUpdated = true; return true;
            }
        }
        // The whole linear solve as one persistent launch with the loop state on chip, if the kernel set has one and the problem fits (iw_onchip.h);
        // it ends with PCGLinearUpdate.  A traced solve gets its per-iteration scalars from the kernel (beta numerator by expansion, as below).
        if (!distributed && (preArg || E->onChipWithoutPreconditioner()) && onChipAllowed() && sp.lIterations > 0 && takeLease()) {
            double* tr = nullptr;
            if (traceEnabled) {
                if (onChipTraceCap < sp.lIterations) { if (onChipTrace) HIP_CHECK(hipFree(onChipTrace)); onChipTraceCap = sp.lIterations; HIP_CHECK(hipMalloc((void**)&onChipTrace, sizeof(double) * 4 * onChipTraceCap)); }
                tr = onChipTrace;
            }
            if (E->pcgSolveOnChip(r, p, delta, sp.lIterations, tr, nullptr, ctx)) {
                usedOnChip = true; unknownsUpdated = E->onChipAppliedUpdate();
                if (traceEnabled) {
                    std::vector<double> h(4 * (size_t)sp.lIterations);
                    HIP_CHECK(hipMemcpyAsync(h.data(), onChipTrace, sizeof(double) * h.size(), hipMemcpyDeviceToHost, stream));
                    HIP_CHECK(hipStreamSynchronize(stream));
                    for (int k = 0; k < sp.lIterations; ++k) {
                        const double aNum = h[4 * k], aDen = h[4 * k + 1], s2 = h[4 * k + 2], s3 = h[4 * k + 3];
                        const T al = ((T)aDen > T(0)) ? (T)aNum / (T)aDen : T(0);
                        const double bNum = std::fmax(aNum - 2.0 * (double)al * s2 + (double)al * (double)al * s3, 0.0);
                        trace.insert(trace.end(), {(double)sp.nIter, (double)k, aNum, aDen, bNum, 0.0});
                    }
                }
                return true;
            }
            dropLease();
        }
        Reduction prev[4] = {redC, Reduction{}, Reduction{}, Reduction{}};   // alphaNum_0 = sum r.p from PCGInit1
        if (distributed) {   // ghost rows of r_0, M and p_0 (written as 0 by evalJTF / PCGInit1_Finish) come from the slab neighbours once
            exchangeVector(r); exchangeVector(p); if (preArg) exchangeVector(preconditioner);
            prev[0] = forConsumers(redC, 0);
        }
        int cur = 0;
        OptAmd_MailRef mail{nullptr, 0, 0, 0, 0, nullptr};      // where the next launch finds the previous launch's sums if they were posted, not reduced
        for (int lIter = 0; lIter < sp.lIterations; ++lIter) {
            PcgIterArgs<T> a{};
            a.rOld = r; a.ApOld = Ap_X; a.pOld = p; a.rNew = r2; a.ApNew = Ap2; a.pNew = p2; a.delta = delta; a.pre = preArg; a.first = lIter == 0;
            a.aNumPrev = prev[0]; a.aDenPrev = prev[1]; a.s2Prev = prev[2]; a.s3Prev = prev[3];
            a.aNum = &setS[cur][0]; a.aDen = &setS[cur][1]; a.s2 = &setS[cur][2]; a.s3 = &setS[cur][3];
            a.mail = mail;
            // the all-reduce of THIS launch's sums, carried out by the launch itself if the communicator can plan it and the kernel set can post
            bool planned = false;
            OptAmd_MailRef nextMail{nullptr, 0, 0, 0, 0, nullptr};
            if (distributed && commExt.allReducePlan && lIter + 1 < sp.lIterations && !traceEnabled && E->iterPostsItself(false))
                planned = commExt.allReducePlan(comm.ctx, 4, &a.post, &nextMail) != 0;
            if (distributed && lIter > 0 && !E->iterStateExchange) exchangeVector(Ap_X);   // kernel with Ap in memory: r and p ghost rows are kept current by the kernel itself
            if (!distributed && !traceEnabled && trialMode && trialPhase != 2 && nPad * sizeof(T) >= ((size_t)64 << 20) && E->deltaMovable()) { deltaTrial(lIter); a.delta = delta; }
            if (!E->pcgIteration(a, ctx)) { if (lIter == 0) return false; fprintf(stderr, "pcgIteration refused mid-loop\n"); exit(1); }
            std::swap(r, r2); std::swap(Ap_X, Ap2); std::swap(p, p2);
            if (distributed && E->iterStateExchange && E->iterExchangeDue) {   // Ap-free kernel: the neighbours' edge rows of r_k and p_k, one grouped exchange
                std::vector<T*> bases;
                T* vecs[4] = {r, p, nullptr, nullptr};
                int nv = E->iterExchangeVectors(vecs);
                if (nv == 0) nv = 2;
                for (int v = 0; v < nv; ++v) for (size_t i = 0; i < E->unknowns.size(); ++i) bases.push_back(vecs[v] + E->unknowns[i].offset);
                exchangeRows(bases);
            }
            for (int i = 0; i < 4; ++i) prev[i] = setS[cur][i];
            mail = OptAmd_MailRef{nullptr, 0, 0, 0, 0, nullptr};
            if (distributed) {   // one all-reduce of the four sums (ping-pong destination, like the partial sets it replaces)
                // If the communicator can POST it and the next launch's prologue can poll the mailbox, nothing waits between the two launches: the
                // contributions cross the links while the next kernel is being launched and requests its first rows.  The last iteration's sums are needed
                // by the flat kernel that closes the loop: those take the complete all-reduce.
                bool posted = planned;
                if (planned) mail = nextMail;
                if (!posted && commExt.allReducePost && E->iterTakesMail && lIter + 1 < sp.lIterations && !traceEnabled) {
                    const double* ps[4]; int ns[4];
                    for (int i = 0; i < 4; ++i) { ps[i] = setS[cur][i].partials; ns[i] = setS[cur][i].n; }
                    posted = commExt.allReducePost(comm.ctx, ps, ns, 4, &mail, (void*)stream) != 0;
                }
                if (!posted) {
                    double* tot = scal4[cur];
                    reduceAcross(setS[cur], 4, tot);
                    for (int i = 0; i < 4; ++i) { prev[i].partials = tot + i; prev[i].n = 1; }
                }
            }
            if (traceEnabled) {
                const double aNum = hostSumLocal(prev[0]), aDen = hostSumLocal(prev[1]), s2 = hostSumLocal(prev[2]), s3 = hostSumLocal(prev[3]);
                const T al = ((T)aDen > T(0)) ? (T)aNum / (T)aDen : T(0);
                // an energy that does not precondition starts from p_0 = r_0 / 4 (guardedInvert(1)) but continues with z = r, so the
                // first alphaNumerator is a quarter of sum r_0^2 -- which is what the expansion needs (see march_pcgIter, stencil_march.h)
                const double rr = (lIter == 0 && !preArg && !E->usesGraph) ? 4.0 * aNum : aNum;
                const double bNum = std::fmax(rr - 2.0 * (double)al * s2 + (double)al * (double)al * s3, 0.0);
                trace.insert(trace.end(), {(double)sp.nIter, (double)lIter, aNum, aDen, bNum, 0.0});
            }
            cur ^= 1;
        }
        // the last iteration's delta += alpha p (PCGStep2, solver.t:461-462); r, z, p of that iteration are dead.  A kernel set may fold it, its own
        // deferred terms and PCGLinearUpdate into one pass over the unknowns (EnergyOps::finishUpdate).
        if (!distributed && !traceEnabled && sp.lIterations > 0 && E->finishUpdate(p2, p, delta, prev[0], prev[1], ctx)) { unknownsUpdated = true; return true; }
        const T* pLast = E->pcgFinish(p2, delta, ctx);
        if (!pLast) pLast = p;
        finalizeLocal(prev[0], scal + 2);
        { ScopedKernel k(ctx, "PCGStep2_delta"); k_step2FirstHalf<T><<<streamGrid, kBlock, 0, stream>>>(delta, delta, pLast, nPacks, scal + 2, nullptr, 0, prev[1].partials, prev[1].n); }
        return true;
    }
    // ---- the same for Levenberg-Marquardt (energy.h PcgIterArgs, LM fields).  Launch k applies Step2 and Step3 of iteration k-1 and
    // delivers Q_{k-1}, so the q early-out of iteration k-1 (solver.t:1093-1102) is decided after launch k -- by which time the
    // state is exactly the reference's at its break (Step3 runs before fetchQ there too).  Every residual_reset_period-th iteration
    // ends with the reference's split Step2 (delta, A delta, r = b - A delta; :1077-1083) on the generic kernels; the next launch then
    // restarts from that r with beta given directly.  Returns false (nothing touched) if the energy has no such kernel.
    bool runSingleKernelLoopLM(const T* preArg, T Q0, T q_tolerance) {
        if (distributed || traceEnabled || keepReferenceP) return false;
        // The whole LM linear solve as one persistent launch (iw_onchip.h, LMV): CtC, the q early-out and the split residual reset happen on chip, the host
        // sees only delta -- and, for a listening caller (verbosity > 0), the iteration and zeta of the early-out in a pinned word, from which the reference's "breaking at
        // iteration" message is printed once the step has drained (round 6: verbose and silent runs take the SAME path).  Not reproduced there: the message of
        // an early-out decided after the LAST iteration (its test is dead -- the loop has ended -- and the on-chip kernels do not form it).
        if (onChipAllowed() && (preArg || E->onChipWithoutPreconditioner()) && sp.lIterations > 0 && Q0 == T(0) && takeLease()) {
            if (!lmBreak) { HIP_CHECK(hipHostMalloc((void**)&lmBreak, 64)); }
            lmBreak[0] = 0.0; lmBreak[1] = 0.0;
            OnChipLm<T> la{trust_region_radius, min_lm_diagonal, max_lm_diagonal, q_tolerance, sp.residual_reset_period, CtC};
            la.breakInfo = verbosity > 0 ? lmBreak : nullptr;
            if (E->pcgSolveOnChip(r, p, delta, sp.lIterations, nullptr, &la, ctx)) { usedOnChip = true; return true; }
            dropLease();
        }
        if (!delta2) delta2 = allocVec();                       // zero-filled like delta; every launch that updates delta rewrites all of it
        Reduction prev[4] = {redC, Reduction{}, Reduction{}, Reduction{}};
        int cur = 0;
        bool afterReset = false, deltaOwed = false, issued = false, issuedRestart = false;
        Reduction bNumDirect{}, bDenDirect{};
        double qDirect = 0;                                         // Q as the last split residual reset summed it (PcgIterArgs::qInit of the restart launch)
        unsigned tagOf[2] = {0, 0};                                 // tag of the Q partials in redQ / redQ2
        auto pNow = [&]() -> const T* { const T* own = E->iterCurrentP(); return own ? own : p; };      // p of the launch adopted last (a kernel set may keep the directions in buffers of its own)
        // One launch from the current state into the alternate buffers (r2, p2, delta2, setS[cur]); adopted later by pointer swaps.
        auto issue = [&](int k, bool restart) -> bool {
            PcgIterArgs<T> a{};
            a.rOld = r; a.ApOld = Ap_X; a.pOld = p; a.rNew = r2; a.ApNew = Ap2; a.pNew = p2; a.delta = delta; a.deltaOut = delta2; a.pre = preArg; a.first = k == 0;
            a.aNumPrev = prev[0]; a.aDenPrev = prev[1]; a.s2Prev = prev[2]; a.s3Prev = prev[3];
            a.aNum = &setS[cur][0]; a.aDen = &setS[cur][1]; a.s2 = &setS[cur][2]; a.s3 = &setS[cur][3];
            a.CtC = CtC; a.b = b; a.q = (k & 1) ? &redQ2 : &redQ; a.afterReset = restart ? 1 : 0; a.betaNum = bNumDirect; a.betaDen = bDenDirect; a.qInit = qDirect;
            if (taggedQ) { if (++launchTag == 0) ++launchTag; a.qTag = tagOf[k & 1] = launchTag; } else tagOf[k & 1] = 0;      // (0: this launch writes plain partials)
            a.lmRadius = trust_region_radius; a.lmMinDiag = min_lm_diagonal; a.lmMaxDiag = max_lm_diagonal;
            issuedRestart = restart;
            return E->pcgIteration(a, ctx);
        };
        for (int lIter = 0; lIter < sp.lIterations; ++lIter) {
            if (!issued && !issue(lIter, afterReset)) { if (lIter == 0) return false; fprintf(stderr, "pcgIteration refused mid-loop\n"); exit(1); }
            issued = false;
            // adopt launch lIter
            const bool appliedStep2 = lIter > 0 && !issuedRestart;    // it finished iteration lIter-1 (delta, r, z, p) and summed Q_{lIter-1}
            std::swap(r, r2); std::swap(p, p2); std::swap(Ap_X, Ap2);
            if (appliedStep2 && E->iterWroteDelta()) std::swap(delta, delta2);      // (a kernel set that pairs its delta updates writes every second launch; what a deferring launch owes: flushOwed)
            bool owedFlushed = false;
            auto flushOwed = [&](int issuedBeyond) { if (!owedFlushed && appliedStep2) { E->iterFlushDelta(delta, issuedBeyond, ctx); owedFlushed = true; } };
            for (int i = 0; i < 4; ++i) prev[i] = setS[cur][i];
            cur ^= 1;
            afterReset = false;
            const bool resetNow = sp.residual_reset_period > 0 && ((lIter + 1) % sp.residual_reset_period) == 0;      // (a period of 0: never -- Lua's x % 0 is nan, solver.t:1077)
            // The split residual reset (solver.t:1077-1083) of this iteration: delta += alpha p, then r = b - (J^T J + CtC) delta afresh.
            // After the last iteration only delta survives (r, z, the beta numerator and -- unless someone listens for the message -- Q are dead), so the
            // reference's computeAdelta and second half are not run then.
            const bool lastAndSilent = lIter + 1 >= sp.lIterations && verbosity == 0;
            auto resetKernels = [&](T* deltaOut) {
                // (a term the adopted launch left owed goes in first, in the same pass; `delta` itself stays as it is: an early-out decided below flushes it there)
                const T *pOwed = nullptr, *aOwed = nullptr;
                if (appliedStep2 && !owedFlushed) (void)E->iterOwedTerm(0, &pOwed, &aOwed);
                { ScopedKernel k(ctx, "PCGStep2_1stHalf");
                  k_step2FirstHalf<T><<<streamGrid, kBlock, 0, stream>>>(delta, deltaOut, pNow(), nPacks, nullptr, prev[0].partials, prev[0].n, prev[1].partials, prev[1].n, pOwed, aOwed); }
                if (lastAndSilent) return;
                if (E->applyJTJResetLM(deltaOut, r, b, preArg, z, CtC, redB, redQR, ctx)) return;      // computeAdelta + the second half in one pass
                E->applyJTJ(deltaOut, Adelta, CtC, nullptr, ctx);             // computeAdelta
                { ScopedKernel k(ctx, "PCGStep2_2ndHalf");
                  k_step2SecondHalf<T><<<streamGrid, kBlock, 0, stream>>>(deltaOut, r, Adelta, b, preArg, z, nPacks, redB.partials, redQR.partials); }
                redB.n = streamGrid; redQR.n = streamGrid;
            };
            bool resetIssued = false;
            if (appliedStep2) {
                const bool tagged = tagOf[lIter & 1] != 0;      // how THIS launch wrote its Q partials (taggedQ may have been switched off since it was issued)
                if (!tagged) beginHostSum((lIter & 1) ? redQ2 : redQ);
                // What follows is enqueued before Q is known.  The next launch writes only the alternate buffers, and the reset writes delta2, r (dead after
                // an early-out) and scratch: if the test below ends the linear solve, their results are simply never adopted (the fetchQ of solver.t:1098
                // no longer idles the GPU).
                if (resetNow) { resetKernels(delta2); resetIssued = true; }
                else if (lIter + 1 < sp.lIterations) { if (!issue(lIter + 1, false)) { fprintf(stderr, "pcgIteration refused mid-loop\n"); exit(1); } issued = true; }
                const T Q1 = (T)(tagged ? pollTaggedSum((lIter & 1) ? redQ2 : redQ, tagOf[lIter & 1]) : endHostSum());
                if (Q1 != Q1) {      // a tagged Q partial never arrived (pollTaggedSum said why): this iteration's test is skipped, Q0 stays the last known value, and the
                    taggedQ = false; //  plan reads Q through the stream (beginHostSum / endHostSum) from here on, so the later early-out tests are real again
                } else {
                    const T zeta = T(lIter) * (Q1 - Q0) / Q1;
                    if (zeta < q_tolerance) {
                        if (verbosity > 0) printf("zeta=%.18g, breaking at iteration: %d\n", (double)zeta, lIter);
                        flushOwed(issued ? 1 : 0);
                        return true;
                    }
                    Q0 = Q1;
                }
            }
            deltaOwed = true;                                          // iteration lIter: Step1 done, its Step2 still to come
            if (resetNow) {
                if (resetIssued) std::swap(delta, delta2); else resetKernels(delta);
                deltaOwed = false;
                // fetchQ (solver.t:1098).  After the last iteration its only effect is the message below: the loop ends either way, so the blocking
                // read (one drain of the stream per outer iteration when residual_reset_period == lIterations, the default) happens only when someone is listening.
                if (!lastAndSilent) {
                    qDirect = hostSum(redQR);
                    const T Q1 = (T)qDirect;
                    const T zeta = T(lIter + 1) * (Q1 - Q0) / Q1;
                    if (zeta < q_tolerance) { if (verbosity > 0) printf("zeta=%.18g, breaking at iteration: %d\n", (double)zeta, lIter + 1); return true; }
                    Q0 = Q1;
                }
                afterReset = true; bNumDirect = redB; bDenDirect = prev[0];
            }
        }
        if (deltaOwed) {   // the last iteration's delta += alpha p; its r, z, p and Q are dead (the reference's last fetchQ can only break a finished loop)
            const T *pOwed = nullptr, *aOwed = nullptr;
            (void)E->iterOwedTerm(0, &pOwed, &aOwed);      // (... behind the term a deferring last launch left owed)
            ScopedKernel k(ctx, "PCGStep2_delta");
            k_step2FirstHalf<T><<<streamGrid, kBlock, 0, stream>>>(delta, delta, pNow(), nPacks, nullptr, prev[0].partials, prev[0].n, prev[1].partials, prev[1].n, pOwed, aOwed);
        }
        return true;
    }
    double hostSumLocal(const Reduction& R) {   // host value of an (already all-reduced, if distributed) reduction
        HIP_CHECK(hipMemcpyAsync(hostBuf, R.partials, R.n * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        double s = 0; for (int i = 0; i < R.n; ++i) s += hostBuf[i];
        return s;
    }
    void finalizeLocal(const Reduction& R, double* dst) { k_finalizeSum<<<1, kBlock, 0, stream>>>(R.partials, R.n, dst); }

    // ---- init (solver.t:956-1007) ---------------------------------------------------------------------
    void init(void** params) override {
        timer.reset(); trace.clear();
        if (overallOpen) { timer.pool.push_back(overallStart); overallOpen = false; }
        if (timer.enabled) { overallStart = timer.get(); HIP_CHECK(hipEventRecord(overallStart, stream)); overallOpen = true; }   // "overall": init -> cleanup (solver.t:959, 1011)
        E->bind(params, ctx);
        boundForSolve = insideSolve;
        sp.nIter = 0; patchSweep = 0;
        if (lm) {
            trust_region_radius = (T)sp.trust_region_radius; radius_decrease_factor = (T)sp.radius_decrease_factor;
            min_lm_diagonal = (T)sp.min_lm_diagonal; max_lm_diagonal = (T)sp.max_lm_diagonal;
        }
        exchangeUnknowns();
        E->precompute(ctx);
        // (inside Opt_ProblemSolve the first step follows at once on the same unknowns: its PCGInit1 rides on this cost pass where the kernel set can -- see stepOnce)
        jtfReady = false; pendingSteps.clear(); E->onChipStepSlot(-1);
        if (!lm && !distributed && insideSolve && E->bindInvariantDuringSolve() && sp.nIterations > 0 && singleKernelAllowed() && r2 && sp.lIterations > 0 &&
            E->evalCostAndJTFInit(redCH, r, p, delta, nPad, redC, ctx)) { jtfReady = true; prevCost = (T)hostSum(redCH); }
        else prevCost = computeCost();
    }
    void cleanup() {   // solver.t:1009-1014
        if (verbosity > 0) printf("final cost=%f\n", (double)prevCost);
        if (timer.enabled) {
            if (overallOpen) { timer.closeRun(stream); KernelTimer::Rec rec{"overall", overallStart, timer.get()}; HIP_CHECK(hipEventRecord(rec.b, stream)); timer.recs.push_back(rec); overallOpen = false; }
            timer.evaluate();
            if (verbosity > 0) timer.print();
        }
    }

    // ---- step of the block-local solver (control flow of the reference comparator, PatchSolverWarping.cu:211-241) ----
    // nIterations outer steps of lIterations sweeps; sweep s shifts the patch tiling by the s-th point (mod 8) of the Halton sequence in bases
    // (2, 3), scaled by the patch size (:208-209).  Each sweep re-linearises at the current unknowns inside its kernel, so there is no global
    // r / p / delta and no grid-wide sum in the loop: one launch per sweep, the cost once per outer step.
    static double radicalInverse(int i, int base) { double f = 1, r = 0; while (i > 0) { f /= base; r += f * (i % base); i /= base; } return r; }
    int stepPatch(void** params) {
        E->bind(params, ctx);
        if (sp.nIter >= sp.nIterations) { cleanup(); return 0; }
        for (int lIter = 0; lIter < sp.lIterations; ++lIter) {
            const int o = patchSweep % 8;
            if (!E->patchIteration((float)radicalInverse(o, 2), (float)radicalInverse(o, 3), sp.patchIterations, sp.patchSize, ctx)) {
                fprintf(stderr, "patchGaussNewtonGPU: unsupported patchSize %d (16 or 32)\n", sp.patchSize); exit(1);
            }
            ++patchSweep;
        }
        E->patchFinish(ctx);
        E->precompute(ctx);
        const T newCost = computeCost();
        if (verbosity > 0) printf("cost: %f -> %f\n", (double)prevCost, (double)newCost);
        prevCost = newCost;
        sp.nIter += 1;
        return 1;
    }

    // ---- step (solver.t:1016-1177) --------------------------------------------------------------------
    // A step whose on-chip linear solve failed on an energy without a launch-per-iteration loop of its own is run again from its PCGInit1 on the generic kernels (the on-chip
    // path is off by then): a loop, not a recursion.
    int step(void** params) override {
        if (patch) return stepPatch(params);
        for (int attempt = 0;; ++attempt) {
            bool again = false;
            const int rc = stepOnce(params, again);
            if (!again || attempt >= 1) return rc;
        }
    }
    // Behind a drain: the costs of the deferred steps in order (printed as the reference prints them), and the first step whose on-chip solve gave up, or -1.
    int settlePending() {
        int failed = -1;
        for (const PendingStep& ps : pendingSteps) {
            if (ps.onChip && E->onChipStepFailed(ps.slot)) { failed = ps.step; break; }
            double c = 0; for (int i = 0; i < costRing[ps.slot].n; ++i) c += costRing[ps.slot].partials[i];
            if (verbosity > 0) printf("cost: %f -> %f\n", (double)prevCost, (double)(T)c);
            prevCost = (T)c; lastStepOnChip = ps.onChip;
        }
        pendingSteps.clear();
        E->onChipClearStepSlots();
        return failed;
    }
    // The deferred step `f` found its on-chip solve timed out: nothing has been applied from it on (the flag is sticky: later launches return at once and every guarded
    // update is skipped), so the solve goes back to step f on the streaming kernels.  r, p of the newest cost + PCGInit1 pass belong to exactly those unknowns.
    void rewindTo(int f) {
        (void)E->onChipFailed();      // (consumes the launch's failure word)
        dropLease();
        ++onChipFailures; onChipCleanSteps = 0; onChipBackoff = std::min(8 << std::min(onChipFailures - 1, 7), 1024);
        if (onChipFailures <= 3)
            fprintf(stderr, "Opt(amd): a wait inside the on-chip PCG kernel timed out (its workgroups were not co-resident: is the GPU shared?); the solve goes back to that step "
                            "with the streaming kernels and the plan stays on them for %d steps before it tries the chip again\n", onChipBackoff);
        onChipOk = false; onChipFellBack = true; lastStepOnChip = false; usedOnChip = false;
        sp.nIter = f;
    }
    int stepOnce(void** params, bool& again) {
        const T min_relative_decrease = (T)sp.min_relative_decrease, min_trust_region_radius = (T)sp.min_trust_region_radius;
        const T max_trust_region_radius = (T)sp.max_trust_region_radius, q_tolerance = (T)sp.q_tolerance, function_tolerance = (T)sp.function_tolerance;
        T Q0 = 0, Q1 = 0;
        if (!(insideSolve && boundForSolve && E->bindInvariantDuringSolve())) E->bind(params, ctx);      // (EnergyOps::bindInvariantDuringSolve: once per Opt_ProblemSolve where nothing bind() derives can have changed)
        const bool mayDefer = !lm && !distributed && insideSolve && !traceEnabled && E->supportsDeferredSteps();
        if (!pendingSteps.empty() && (!mayDefer || sp.nIter >= sp.nIterations)) {      // (cannot happen: the last step of a solve is never deferred -- kept so that no cost is ever lost)
            drain();
            const int f = settlePending();
            if (f >= 0) { rewindTo(f); jtfReady = false; }
        }
        if (sp.nIter >= sp.nIterations) { cleanup(); return 0; }
        const int deferSlot = mayDefer ? (int)pendingSteps.size() : -1;
        E->onChipStepSlot(deferSlot);
        bool deferredNow = false;
        const T* preArg = E->usePreconditioner ? preconditioner : nullptr;   // solver.t:467-470: pre = 1 unless the energy preconditions

        // PCGInit1 [+ _Graph + _Finish]: the energy produces r = -J^T F and raw diag(J^T J) (parked in CtC) -- or, for the Gauss-Newton single-kernel loop on
        // one GPU, r, p = M r, delta = 0 and the partial sums of r.p directly (EnergyOps::evalJTFInit)
        unknownsUpdated = false;
        const bool jtfCarried = jtfReady && insideSolve;      // r, p and the partial sums of r.p are already there: the previous step's cost pass wrote them from these very unknowns
        jtfReady = false;
        const bool fusedInit = !lm && !distributed && singleKernelAllowed() && r2 && sp.lIterations > 0 && (jtfCarried || E->evalJTFInit(r, p, delta, nPad, redC, ctx));
        bool fusedInitLM = false;
        if (lm && !distributed) {
            LmInitArgs<T> la{CtC, SSq, r, delta, preconditioner, b, p, trust_region_radius, min_lm_diagonal, max_lm_diagonal, sp.nIter == 0 ? 1 : 0, &redC, &redQ};
            fusedInitLM = E->evalJTFInitLM(la, ctx);
        }
        if (!fusedInit && !fusedInitLM) E->evalJTF(r, CtC, ctx);
        if (!lm && !fusedInit) {
            ScopedKernel k(ctx, "PCGInit1_Finish");
            k_initFinish<T><<<streamGrid, kBlock, 0, stream>>>(r, CtC, preconditioner, p, delta, nPacks, E->usePreconditioner ? 1 : 0, E->usesGraph ? 1 : 0, redC.partials);
            redC.n = streamGrid;
        }
        aSlot = 0;
        if (lm) {
            if (!fusedInitLM) {   // PCGInit1_Finish + (first outer iteration) PCGSaveSSq + PCGFinalizeDiagonal in one pass
                ScopedKernel k(ctx, "PCGFinalizeDiagonal");
                k_finalizeDiagonal<T, true><<<streamGrid, kBlock, 0, stream>>>(CtC, SSq, r, delta, preconditioner, b, p, nPacks, trust_region_radius, min_lm_diagonal,
                                                                               max_lm_diagonal, redC.partials, redQ.partials, E->usePreconditioner ? 1 : 0,
                                                                               E->usesGraph ? 1 : 0, sp.nIter == 0 ? 1 : 0);
                redC.n = streamGrid; redQ.n = streamGrid;
            }
            preArg = E->usePreconditioner ? preconditioner : nullptr;
            // fetchQ, solver.t:1050: Q_0 = 1/2 sum delta . (r + b) with the delta PCGInit1 has just zeroed -- exactly 0 for every finite r, so the
            // blocking read (one full drain of the stream per outer iteration) is not performed
            Q0 = T(0);
        }

        // Loop structure: the reference runs Step1, Step2, Step3 per iteration (:1056-1103).  Here Step3 of
        // iteration k is fused into Step1 of iteration k+1 when the energy offers that kernel (it only
        // feeds the next Step1; after the last iteration p is dead).
        bool pendingStep3 = false;
        Reduction bNum;
        const bool single = singleKernelAllowed() && r2 && (lm ? (oneKernelLM && runSingleKernelLoopLM(preArg, Q0, q_tolerance)) : runSingleKernelLoop(preArg));
        if (fusedInit && !single) {      // the kernel set accepted evalJTFInit but refused the loop (it should not): PCGInit1 again for the generic loop -- a library does not exit()
            fprintf(stderr, "Opt(amd): the kernel set accepted evalJTFInit but refused the single-kernel loop; this step runs on the generic kernels\n");
            E->evalJTF(r, CtC, ctx);
            ScopedKernel k(ctx, "PCGInit1_Finish");
            k_initFinish<T><<<streamGrid, kBlock, 0, stream>>>(r, CtC, preconditioner, p, delta, nPacks, E->usePreconditioner ? 1 : 0, E->usesGraph ? 1 : 0, redC.partials);
            redC.n = streamGrid;
        }
        if (!single) finalizeTo(redC, scal + aSlot);   // alphaNumerator = sum r.p as one device scalar (the single-kernel loops sum the partials in their first launch)
        // Step3 of the previous iteration (when pending) and Step1 of the next one.  None of it touches what survives a q early-out
        // (delta, and p only through the very Step3 the reference also runs before its q test), so in LM it is enqueued BEFORE the
        // host reads q of the current iteration: the blocking fetchQ of solver.t:1098 then overlaps with useful kernels instead of
        // draining the GPU once per PCG iteration.  With tracing on, the original order (decide, then launch) is kept.
        Reduction aDen;
        auto stepThreeAndOne = [&]() {
            bool applied = false;
            if (pendingStep3) {
                exchangeVector(z);      // (an energy without the fused kernel refuses: the generic PCGStep3 below)
                applied = E->applyJTJFused(p, z, p2, Ap_X, lm ? CtC : nullptr, &redA, bNum, scal + aSlot, scal + (aSlot ^ 1), ctx);
                if (applied) std::swap(p, p2);
                if (!applied) {
                    ScopedKernel k(ctx, "PCGStep3");
                    k_step3<T><<<streamGrid, kBlock, 0, stream>>>(z, p, nPacks, bNum.partials, bNum.n, scal + aSlot, scal + (aSlot ^ 1));
                }
                aSlot ^= 1;   // alphaNumerator <- betaNumerator (solver.t:1091)
                pendingStep3 = false;
            }
            if (!applied) {
                exchangeVector(p);
                E->applyJTJ(p, Ap_X, lm ? CtC : nullptr, &redA, ctx);    // PCGStep1 (+_Graph)
            }
            aDen = forConsumers(redA, 0);
        };
        const bool speculate = !traceEnabled;
        if (!single && sp.lIterations > 0) stepThreeAndOne();     // Step1 of iteration 0
        for (int lIter = 0; !single && lIter < sp.lIterations; ++lIter) {
            const bool reset = lm && sp.residual_reset_period > 0 && ((lIter + 1) % sp.residual_reset_period) == 0;
            // After the last iteration only delta survives: r, z, the beta numerator and (unless someone listens for the "breaking" message) Q are dead, so the
            // last PCGStep2 -- or the last split residual reset -- shrinks to its delta += alpha p.
            const bool deltaOnly = lIter + 1 >= sp.lIterations && !traceEnabled && !keepReferenceP && (!lm || verbosity == 0);
            if (deltaOnly) {
                ScopedKernel k(ctx, "PCGStep2_delta");
                k_step2FirstHalf<T><<<streamGrid, kBlock, 0, stream>>>(delta, delta, p, nPacks, scal + aSlot, nullptr, 0, aDen.partials, aDen.n);
                break;
            }
            if (reset) {   // solver.t:1077-1083
                { ScopedKernel k(ctx, "PCGStep2_1stHalf"); k_step2FirstHalf<T><<<streamGrid, kBlock, 0, stream>>>(delta, delta, p, nPacks, scal + aSlot, nullptr, 0, aDen.partials, aDen.n); }
                exchangeVector(delta);
                E->applyJTJ(delta, Adelta, CtC, nullptr, ctx);             // computeAdelta (+_Graph)
                { ScopedKernel k(ctx, "PCGStep2_2ndHalf");
                  k_step2SecondHalf<T><<<streamGrid, kBlock, 0, stream>>>(delta, r, Adelta, b, preArg, z, nPacks, redB.partials, redQ.partials); }
                redB.n = streamGrid; redQ.n = streamGrid;
            } else {
                ScopedKernel k(ctx, "PCGStep2");
                if (lm) k_step2<T, true><<<streamGrid, kBlock, 0, stream>>>(delta, p, r, Ap_X, preArg, b, z, nPacks, scal + aSlot, aDen.partials, aDen.n, redB.partials, redQ.partials);
                else k_step2<T, false><<<streamGrid, kBlock, 0, stream>>>(delta, p, r, Ap_X, preArg, nullptr, z, nPacks, scal + aSlot, aDen.partials, aDen.n, redB.partials, nullptr);
                redB.n = streamGrid; redQ.n = streamGrid;
            }
            bNum = forConsumers(redB, 1);
            pendingStep3 = true;   // PCGStep3 of this iteration runs with the next PCGStep1
            const bool more = lIter + 1 < sp.lIterations;
            double qh = 0;
            const bool deadFetch = lm && !more && verbosity == 0 && !traceEnabled;   // fetchQ after the last iteration only feeds the "breaking" message
            if (deadFetch) {
            } else if (lm && speculate) {
                beginHostSum(redQ);
                if (more) stepThreeAndOne();
                qh = endHostSum();
            } else {
                if (lm) qh = hostSum(redQ);
                if (traceEnabled) record(lIter, aDen, bNum, qh);
            }
            if (lm && !deadFetch) {   // solver.t:1093-1102
                Q1 = (T)qh;
                T zeta = T(lIter + 1) * (Q1 - Q0) / Q1;
                if (zeta < q_tolerance) { if (verbosity > 0) printf("zeta=%.18g, breaking at iteration: %d\n", (double)zeta, lIter + 1); break; }
                Q0 = Q1;
            }
            if (more && !(lm && speculate)) stepThreeAndOne();
        }
        if (pendingStep3 && keepReferenceP) {   // the reference's final PCGStep3 only matters to someone probing `p`
            ScopedKernel k(ctx, "PCGStep3");
            k_step3<T><<<streamGrid, kBlock, 0, stream>>>(z, p, nPacks, bNum.partials, bNum.n, scal + aSlot, scal + (aSlot ^ 1));
            aSlot ^= 1;
        }

        T model_cost_change = 0;
        T newCost = 0;
        // What follows the linear solve (solver.t:1108-1117): model cost, savePreviousUnknowns + PCGLinearUpdate, precompute, the new cost.  The reference reads the
        // model cost, then updates, then reads the new cost (two blocking copies); neither value steers anything before both are known, so all of it is enqueued
        // and the stream is drained once.
        auto afterLinearSolve = [&]() {
            if (lm) {   // solver.t:1108-1113, 819-827
                exchangeVector(delta);
                E->evalModelCost(delta, distributed ? redA : redMH, ctx);   // (its own partials buffer: the value is read together with the new cost below)
                imageOp(3);   // savePreviousUnknowns + PCGLinearUpdate
            } else if (!unknownsUpdated) {   // PCGLinearUpdate (behind an on-chip solve that left the update to the solver: guarded by the launch's failure word where the kernel set can)
                if (!(usedOnChip && E->onChipGuardedUpdate(delta, ctx))) imageOp(0);
            }
            exchangeUnknowns();
            E->precompute(ctx);
            if (lm && !distributed) {
                E->evalCost(redCH, ctx);                   // both sets of partials are written straight to pinned memory: one drain, no copy kernels
                drain();
                if (usedOnChip && verbosity > 0 && lmBreak && lmBreak[0] > 0.0 && !E->onChipFailedPeek()) { printf("zeta=%.18g, breaking at iteration: %d\n", lmBreak[1], (int)lmBreak[0] - 1); lmBreak[0] = 0.0; }
                double sm = 0, sc = 0;
                for (int i = 0; i < redMH.n; ++i) sm += redMH.partials[i];
                for (int i = 0; i < redCH.n; ++i) sc += redCH.partials[i];
                newCost = (T)sc;
                const T model_cost = (T)sm;
                if (verbosity > 0) printf(" cost=%f \n model_cost=%f \n", (double)prevCost, (double)model_cost);
                model_cost_change = prevCost - model_cost;
                if (verbosity > 0) printf(" model_cost_change=%f \n", (double)model_cost_change);
            } else {
                if (lm) {
                    T model_cost = (T)hostSum(redA);
                    if (verbosity > 0) printf(" cost=%f \n model_cost=%f \n", (double)prevCost, (double)model_cost);
                    model_cost_change = prevCost - model_cost;
                    if (verbosity > 0) printf(" model_cost_change=%f \n", (double)model_cost_change);
                }
                // Gauss-Newton inside Opt_ProblemSolve with another step to come: that step's PCGInit1 reads the unknowns this cost reads, and no caller code runs in
                // between -- one pass does both where the kernel set can (same cost bits: same grid, same expressions).  A step on the generic kernels (back-off after
                // an on-chip time-out included) does not ask.
                const bool carry = !lm && !distributed && insideSolve && boundForSolve && E->bindInvariantDuringSolve() && sp.nIter + 1 < sp.nIterations &&
                                   singleKernelAllowed() && r2 && sp.lIterations > 0;
                Reduction& costR = deferSlot >= 0 ? costRing[deferSlot] : redCH;
                if (deferSlot >= 0 && !costR.partials) { HIP_CHECK(hipHostMalloc((void**)&costR.partials, 2 * kMaxPartials * sizeof(double))); costR.hostVisible = true; }
                if (carry && E->evalCostAndJTFInit(costR, r, p, delta, nPad, redC, ctx)) {
                    jtfReady = true;
                    if (deferSlot >= 0 && deferSlot + 1 < kDefer) deferredNow = true;      // nothing is read back now: the next step is enqueued behind this one
                    else newCost = (T)hostSum(costR);
                } else newCost = computeCost();
            }
        };
        afterLinearSolve();

        if (deferredNow) {      // (the lease stays with this plan until the steps are settled)
            pendingSteps.push_back({deferSlot, sp.nIter, usedOnChip});
            usedOnChip = false;
            sp.nIter += 1;
            if (!onChipOk && onChipFailures > 0 && ++onChipCleanSteps >= onChipBackoff) {      // back-off over (these were streaming steps): the next step tries the chip again
                onChipOk = true; onChipFellBack = false;
                E->onChipRearm(ctx);
            }
            return 1;
        }
        if (!pendingSteps.empty()) {      // the stream has drained (this step's cost was read): what the deferred steps before it left
            const bool thisOnChip = usedOnChip;
            const int f = settlePending();
            if (f >= 0) {
                // from step f on nothing was applied -- unless THIS step ran on the streaming kernels (it cannot while the path is armed; if it did, it was a valid step from
                // the unknowns of step f and counts as that step)
                rewindTo(f);
                if (!thisOnChip) { if (verbosity > 0) printf("cost: %f -> %f\n", (double)prevCost, (double)newCost); prevCost = newCost; sp.nIter = f + 1; }
                return 1;      // (jtfReady says whether this step's cost pass carried a PCGInit1 for the unknowns as they stand; the last step of a solve carries none)
            }
            if (!usedOnChip) dropLease();
        }
        lastStepOnChip = usedOnChip;
        if (usedOnChip) {      // (the stream has drained: the cost was read)
            usedOnChip = false;
            bool ocFailedNow = E->onChipFailed();
            dropLease();      // the launch has left the chip
            if (ocFailedNow) {
                ++onChipFailures; onChipCleanSteps = 0; onChipBackoff = std::min(8 << std::min(onChipFailures - 1, 7), 1024);
                if (onChipFailures <= 3)
                    fprintf(stderr, "Opt(amd): a wait inside the on-chip PCG kernel timed out (its workgroups were not co-resident: is the GPU shared?); this linear solve is redone "
                                    "with the streaming kernels and the plan stays on them for %d steps before it tries the chip again\n", onChipBackoff);
                onChipOk = false; onChipFellBack = true; lastStepOnChip = false; unknownsUpdated = false;
                if (lm) {      // the kernel produced no delta: the update above added nothing meaningful -- back to the saved unknowns, then the launch-per-iteration loop
                    imageOp(2);
                    E->precompute(ctx);      // (workgroups that had finished before the others gave up may have written their delta: the update above was then not the identity)
                    HIP_CHECK(hipMemsetAsync(delta, 0, nPad * sizeof(T), stream));
                    // an energy whose LM loop is the generic one (no single-kernel LM iteration): the whole step again, from its PCGInit1, on the generic kernels
                    if (!runSingleKernelLoopLM(preArg, T(0), q_tolerance)) { again = true; return 1; }
                } else {
                    // Gauss-Newton: nothing was applied (iw_applyDelta checks the flag -- in slab mode the all-reduced verdict, so no rank kept its update).
                    // The redone loop starts from delta = 0 as PCGInit1 left it: the ROWS = 16 variant accumulates delta in memory while it runs.
                    HIP_CHECK(hipMemsetAsync(delta, 0, nPad * sizeof(T), stream));
                    if (!runSingleKernelLoop(preArg)) { again = true; return 1; }      // (no single-kernel loop either: the whole step again on the generic kernels)
                }
                afterLinearSolve();
            }
        }
        if (lm) {   // solver.t:1119-1157
            T cost_change = prevCost - newCost;
            T relative_decrease = cost_change / model_cost_change;
            if (cost_change >= 0 && relative_decrease > min_relative_decrease) {
                T absolute_function_tolerance = prevCost * function_tolerance;
                if (cost_change <= absolute_function_tolerance) {
                    if (verbosity > 0) printf("\nFunction tolerance reached, exiting\n");
                    cleanup(); return 0;
                }
                // Terra promotes these literals to double; results are stored back as opt_float (solver.t:1135-1139)
                double step_quality = (double)relative_decrease, min_factor = 1.0 / 3.0;
                double tmp_factor = 1.0 - std::pow(2.0 * step_quality - 1.0, 3.0);
                trust_region_radius = (T)((double)trust_region_radius / std::fmax(min_factor, tmp_factor));
                trust_region_radius = std::fmin(trust_region_radius, max_trust_region_radius);
                radius_decrease_factor = T(2.0);
                prevCost = newCost;
            } else {
                imageOp(2);   // revertUpdate
                trust_region_radius = trust_region_radius / radius_decrease_factor;
                if (verbosity > 0) printf(" trust_region_radius=%f \n", (double)trust_region_radius);
                radius_decrease_factor = T(2.0) * radius_decrease_factor;
                if (trust_region_radius <= min_trust_region_radius) {
                    if (verbosity > 0) printf("\nTrust_region_radius is less than the min, exiting\n");
                    cleanup(); return 0;
                }
                if (verbosity > 0) printf("REVERT\n");
                exchangeUnknowns();
                E->precompute(ctx);
                HIP_CHECK(hipStreamSynchronize(stream));   // results visible when the call returns
            }
        } else {
            if (verbosity > 0) printf("cost: %f -> %f\n", (double)prevCost, (double)newCost);
            prevCost = newCost;
        }
        sp.nIter += 1;
        if (!onChipOk && onChipFailures > 0 && ++onChipCleanSteps >= onChipBackoff) {      // back-off over: the next step tries the chip again
            onChipOk = true; onChipFellBack = false;
            E->onChipRearm(ctx);
        }
        return 1;
    }

    double cost() const override { return (double)prevCost; }   // solver.t:1179-1182
    void setTiming(int mode) override {      // per-kernel hipEvents from the next launch on (what collectPerKernelTimingInfo sets at plan time); 2: one pair per run of equal names
        HIP_CHECK(hipStreamSynchronize(stream));
        if (overallOpen) { timer.pool.push_back(overallStart); overallOpen = false; }
        const bool on = mode != 0;
        timer.reset(); timer.enabled = on; timer.coarse = mode == 2; ctx.timer = on ? &timer : nullptr;
    }
    long numUnknownScalars() const override { return n; }
    double trustRegionRadius() const override { return (double)trust_region_radius; }
    int onChipStatus() const override { return onChipFellBack ? 2 : lastStepOnChip ? 1 : 0; }
    std::string describe() override {
        std::string d = sp.amd_reference_order ? std::string("path=reference-order (PCGStep1 [+ the previous PCGStep3] and PCGStep2 per PCG iteration, r / z / A p in memory); amd_reference_order=1")
                                               : E->describe(sp.lIterations, lm);
        if (!sp.amd_reference_order && !sp.amd_onchip) d += "; amd_onchip=0";
        if (!sp.amd_reference_order && !oneKernel) d += "; OPT_AMD_ONEKERNEL=0 (reference-order loop by environment)";
        if (onChipFailures) d += "; onchip_fallbacks=" + std::to_string(onChipFailures) + "; onchip_backoff_steps_left=" + std::to_string(onChipOk ? 0 : onChipBackoff - onChipCleanSteps);
        // what Opt_ProblemSolve does beyond Init + Step by Step on this plan, and where the trial over delta's placement stands
        if (!lm && !distributed && E->supportsDeferredSteps()) d += "; solve_enqueues_steps_back_to_back=up to " + std::to_string(kDefer - 1);
        if (!lm && !distributed && E->deltaMovable() && nPad * sizeof(T) >= ((size_t)64 << 20))
            d += std::string("; delta_placement_trial=") + (!trialMode ? "off" : trialPhase == 2 ? "done" : trialPhase == 1 ? "running" : "before the first linear solve of more than 28 launches");
        d += std::string("; solver=") + (lm ? "LM" : "GN") + "; distributed=" + (distributed ? "yes" : "no") + "; comm_world=" + std::to_string(distributed ? comm.world : 1) +
             "; comm_ext=" + (commExt.onChipPlan ? "onChipPlan " : "") + (commExt.allReducePost ? "allReducePost " : "") + (commExt.allReducePartials ? "allReducePartials" : "");
        return d;
    }
    void* vector(const std::string& nm) override {
        if (nm == "delta") return delta; if (nm == "r") return r; if (nm == "b") return b; if (nm == "Adelta") return Adelta;
        if (nm == "z") return z; if (nm == "p") return p; if (nm == "Ap_X") return Ap_X; if (nm == "CtC") return CtC;
        if (nm == "preconditioner") return preconditioner; if (nm == "SSq") return SSq; if (nm == "prevX") return prevX;
        return nullptr;
    }

    // ---- kernel-level probes (OptAmd.h) ---------------------------------------------------------------
    void evalJTF(void** params, void* jtf, void* diag) override {
        E->bind(params, ctx); exchangeUnknowns(); E->precompute(ctx);
        E->evalJTF(z, Ap_X, ctx);   // scratch use of z / Ap_X: z = -J^T F
        HIP_CHECK(hipStreamSynchronize(stream));
        std::vector<T> h(n);
        HIP_CHECK(hipMemcpy(h.data(), z, n * sizeof(T), hipMemcpyDeviceToHost));
        for (auto& x : h) x = -x;
        HIP_CHECK(hipMemcpy(jtf, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(diag, Ap_X, n * sizeof(T), hipMemcpyDeviceToDevice));
    }
    double applyJTJ(void** params, const void* v, void* out) override {
        E->bind(params, ctx); exchangeUnknowns(); E->precompute(ctx);
        E->evalJTF(z, CtC, ctx);   // refreshes the energy's per-iteration auxiliaries (e.g. cos/sin tables)
        HIP_CHECK(hipMemsetAsync(p, 0, nPad * sizeof(T), stream));
        HIP_CHECK(hipMemcpyAsync(p, v, n * sizeof(T), hipMemcpyDeviceToDevice, stream));
        exchangeVector(p);
        E->applyJTJ(p, Ap_X, nullptr, &redA, ctx);
        double d = hostSum(redA);
        HIP_CHECK(hipMemcpy(out, Ap_X, n * sizeof(T), hipMemcpyDeviceToDevice));
        return d;
    }
    double evalCost(void** params) override {
        E->bind(params, ctx); exchangeUnknowns(); E->precompute(ctx);
        return (double)computeCost();
    }
    int setSlab(long row0, long rows, long globalHeight, const OptAmd_SlabComm* c) override {
        if (!E->supportsSlab() || !c || patch) return 0;   // the block-local solver is single-GPU
        // local height = rows + 2 * ghost: the plan was created with dims {W, rows + 2 * ghost}
        const long localH = E->unknowns[0].elems * E->unknowns[0].channels / E->rowScalars(0);
        const long g = (localH - rows) / 2;
        if (g < 1 || rows + 2 * g != localH) return 0;
        E->slab.active = true; E->slab.ghost = (int)g; E->slab.yBegin = (int)g; E->slab.yEnd = (int)(rows + g); E->slab.gy0 = (int)(row0 - g); E->slab.Hg = (int)globalHeight;
        comm = *c; commExt = OptAmd_SlabCommExt{};
        distributed = c->world > 1 || getenv("OPT_AMD_FORCE_COMM") != nullptr;   // the env switch lets a 1-rank test drive the comm callbacks
        return 1;
    }
    int setSlabExt(const OptAmd_SlabCommExt* e) override {      // only the members the caller's struct actually has are read
        if (!e || !E->slab.active) return 0;
        commExt = OptAmd_SlabCommExt{};
        const size_t have = (size_t)e->size;
        auto has = [&](size_t off, size_t sz) { return have >= off + sz; };
        if (has(offsetof(OptAmd_SlabCommExt, allReducePartials), sizeof(e->allReducePartials))) commExt.allReducePartials = e->allReducePartials;
        if (has(offsetof(OptAmd_SlabCommExt, allReducePost), sizeof(e->allReducePost))) commExt.allReducePost = e->allReducePost;
        if (has(offsetof(OptAmd_SlabCommExt, allReducePlan), sizeof(e->allReducePlan))) commExt.allReducePlan = e->allReducePlan;
        if (has(offsetof(OptAmd_SlabCommExt, onChipPlan), sizeof(e->onChipPlan))) { commExt.onChipPlan = e->onChipPlan; E->onChipPlan = e->onChipPlan; E->onChipCtx = comm.ctx; }
        commExt.size = sizeof(OptAmd_SlabCommExt);
        return 1;
    }
};

bool SolverBase::setParameter(const char* name, const void* value) {   // solver.t:1205-1221
    std::string nm(name);
#define PF(x) if (nm == #x) { sp.x = *(const float*)value; return true; }
#define PI(x) if (nm == #x) { sp.x = *(const int*)value; return true; }
    PF(min_relative_decrease) PF(min_trust_region_radius) PF(max_trust_region_radius) PF(q_tolerance) PF(function_tolerance)
    PF(trust_region_radius) PF(radius_decrease_factor) PF(min_lm_diagonal) PF(max_lm_diagonal)
    PI(residual_reset_period) PI(nIter) PI(nIterations) PI(lIterations) PI(patchIterations) PI(patchSize)
    PI(amd_reference_order) PI(amd_onchip)
#undef PF
#undef PI
    return false;
}

SolverBase* makeSolver(const EnergyInfo& info, bool lm, bool patch, bool doublePrecision, const unsigned* dims, bool timing, int verbosity) {
    if (doublePrecision && !info.floatOnly) {
        EnergyOps<double>* e = info.makeDouble(dims);
        if (!e) return nullptr;
        if (patch && !e->supportsPatch()) { delete e; return nullptr; }
        auto* s = new PcgSolver<double>(e, lm, timing, verbosity); s->patch = patch; return s;
    }
    EnergyOps<float>* e = info.makeFloat(dims);
    if (!e) return nullptr;
    if (patch && !e->supportsPatch()) { delete e; return nullptr; }
    auto* s = new PcgSolver<float>(e, lm, timing, verbosity); s->patch = patch; return s;
}

}  // namespace optamd
