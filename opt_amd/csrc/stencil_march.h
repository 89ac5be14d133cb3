// Marching PCG iteration for 5-point stencils with C channels per pixel: the scheme of the image_warping streaming kernel (iw_iter.h) as a template over the
// energy's operator, for the hand-written linear image energies (poisson_image_editing: C = 4; tests/minimal laplacian: C = 1).
//
// The reference runs PCGStep1 -> sum -> PCGStep2 -> sum -> PCGStep3 per iteration (solverGPUGaussNewton.t:1056-1092; per-thread J^T J p: o.t:2029-2089).  Launch k
// of march_pcgIter does, for every pixel it touches,  r_k = r_{k-1} - alpha_{k-1} A p_{k-1};  p_k = r_k + beta_{k-1} p_{k-1}  (neither energy preconditions: z = r),
// then  A p_k  on its own rows with the sums  alphaDen = p.Ap,  alphaNum = sum r^2,  s2 = sum r.Ap,  s3 = sum Ap^2  (beta by expansion: energy.h PcgIterArgs).
//   * A p is never stored: the launch reads p_{k-1} on a 2-pixel ring and evaluates the stencil twice (A p_{k-1} again on the 1-ring, A p_k on its own pixels);
//   * there is no residual vector: p_{k-1} = r_{k-1} + beta_{k-2} p_{k-2} determines r_{k-1}, so the loop state is a ring of three p buffers (the first two
//     launches read the solver's r_0);
//   * delta is touched every second launch: delta += alpha_{k-2} p_{k-2} + alpha_{k-1} p_{k-1}, both operands being rows the launch loads anyway.
// Per pixel and iteration: p_{k-1}, p_{k-2} in, p_k out, delta in + out every second launch, one flag byte  =  4 vectors + 1 B  (poisson float: 65 B against
// 100 B of the one-thread-per-pixel kernel it replaces and 212 B of the three-kernel loop).
// Structure (iw_iter.h): a workgroup owns a column strip and a contiguous range of rows; a lane keeps three rows of p_{k-1} and three of p_k of its column in
// registers (trip y turns the freshly loaded row y+2 into A p_{k-1}(y+1), r_k, p_k(y+1), then A p_k(y)); horizontal neighbours are whole-wave DPP shifts (a wave
// covers 64 pixels and produces the inner 60); three raw row buffers are requested three rows ahead and rotated by name; rows are addressed through buffer
// descriptors (soffset = row, voffset = lane constant); successive launches sweep top-down / bottom-up (FLIP) so that a launch starts on the rows the previous
// one left in the caches.
//
// The operator:
//   struct Op { static constexpr int C;  static constexpr bool kMasked;      // channels; does a flag byte per pixel switch pixels off (Exclude, o.t:2452-2455)?
//               static constexpr int kCoef;                                  // per-pixel coefficients of the operator (0: none), streamed from MarchK::coef once per row
//               static constexpr bool kSplit31;                              // the unknown vector is a 3-channel image followed by a 1-channel image (C = 4), see marchLoadSplit
//               __device__ MVec<T, C> apply(pc, pl, pr, pu, pd, hasL, hasR, hasU, hasD, coef) const; }     // (J^T J p) at an active pixel; p of an inactive / absent pixel is 0
#pragma once
#include "iw_device.h"

namespace optamd {
namespace {

template <class T, int C> struct MVec { T v[C]; };
constexpr int kMarchSpan = kWave - 4;      // pixels a wave produces per row (two DPP rings)

template <class T, int C> __device__ __forceinline__ MVec<T, C> marchLoad(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so) {
    MVec<T, C> o;
    constexpr int B = C * (int)sizeof(T);
    static_assert(B == 4 || B == 8 || B == 16 || B == 32, "1, 2 or 4 channels");
    if constexpr (B == 4) { const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(r, (int)v, (int)so, 0); __builtin_memcpy(&o, &w, 4); }
    else if constexpr (B == 8) { const iw_u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)v, (int)so, 0); __builtin_memcpy(&o, &w, 8); }
    else if constexpr (B == 16) { const iw_u4 w = __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, (int)so, 0); __builtin_memcpy(&o, &w, 16); }
    else {
        const iw_u4 w0 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, (int)so, 0), w1 = __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, (int)(so + 16u), 0);
        __builtin_memcpy(&o, &w0, 16); __builtin_memcpy(reinterpret_cast<char*>(&o) + 16, &w1, 16);
    }
    return o;
}
template <class T, int C> __device__ __forceinline__ void marchStore(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so, const MVec<T, C>& o) {
    constexpr int B = C * (int)sizeof(T);
    if constexpr (B == 4) { unsigned w; __builtin_memcpy(&w, &o, 4); __builtin_amdgcn_raw_buffer_store_b32(w, r, (int)v, (int)so, 0); }
    else if constexpr (B == 8) { iw_u2 w; __builtin_memcpy(&w, &o, 8); __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)v, (int)so, 0); }
    else if constexpr (B == 16) { iw_u4 w; __builtin_memcpy(&w, &o, 16); __builtin_amdgcn_raw_buffer_store_b128(w, r, (int)v, (int)so, 0); }
    else {
        iw_u4 w0, w1; __builtin_memcpy(&w0, &o, 16); __builtin_memcpy(&w1, reinterpret_cast<const char*>(&o) + 16, 16);
        __builtin_amdgcn_raw_buffer_store_b128(w0, r, (int)v, (int)so, 0); __builtin_amdgcn_raw_buffer_store_b128(w1, r, (int)v, (int)(so + 16u), 0);
    }
}
// A solver vector of two unknown images -- three channels per pixel followed, after all pixels, by one channel per pixel (intrinsic_image_decomposition: r then s; the
// solver's layout, energy.h) -- seen as four channels per pixel: x3 = x * 3 * sizeof(T), x1 = x * sizeof(T), so3 / so1 the row offsets of the two parts
typedef unsigned int march_u3 __attribute__((ext_vector_type(3)));
template <class T> __device__ __forceinline__ MVec<T, 4> marchLoadSplit(__amdgpu_buffer_rsrc_t r, unsigned x3, unsigned so3, unsigned x1, unsigned so1) {
    MVec<T, 4> o;
    if constexpr (sizeof(T) == 4) {
        const march_u3 a = __builtin_amdgcn_raw_buffer_load_b96(r, (int)x3, (int)so3, 0); const unsigned b = __builtin_amdgcn_raw_buffer_load_b32(r, (int)x1, (int)so1, 0);
        __builtin_memcpy(&o, &a, 12); __builtin_memcpy(reinterpret_cast<char*>(&o) + 12, &b, 4);
    } else {
        const iw_u4 a = __builtin_amdgcn_raw_buffer_load_b128(r, (int)x3, (int)so3, 0); const iw_u2 a2 = __builtin_amdgcn_raw_buffer_load_b64(r, (int)x3, (int)(so3 + 16u), 0);
        const iw_u2 b = __builtin_amdgcn_raw_buffer_load_b64(r, (int)x1, (int)so1, 0);
        __builtin_memcpy(&o, &a, 16); __builtin_memcpy(reinterpret_cast<char*>(&o) + 16, &a2, 8); __builtin_memcpy(reinterpret_cast<char*>(&o) + 24, &b, 8);
    }
    return o;
}
template <class T> __device__ __forceinline__ void marchStoreSplit(__amdgpu_buffer_rsrc_t r, unsigned x3, unsigned so3, unsigned x1, unsigned so1, const MVec<T, 4>& o) {
    if constexpr (sizeof(T) == 4) {
        march_u3 a; unsigned b; __builtin_memcpy(&a, &o, 12); __builtin_memcpy(&b, reinterpret_cast<const char*>(&o) + 12, 4);
        __builtin_amdgcn_raw_buffer_store_b96(a, r, (int)x3, (int)so3, 0); __builtin_amdgcn_raw_buffer_store_b32(b, r, (int)x1, (int)so1, 0);
    } else {
        iw_u4 a; iw_u2 a2, b; __builtin_memcpy(&a, &o, 16); __builtin_memcpy(&a2, reinterpret_cast<const char*>(&o) + 16, 8); __builtin_memcpy(&b, reinterpret_cast<const char*>(&o) + 24, 8);
        __builtin_amdgcn_raw_buffer_store_b128(a, r, (int)x3, (int)so3, 0); __builtin_amdgcn_raw_buffer_store_b64(a2, r, (int)x3, (int)(so3 + 16u), 0);
        __builtin_amdgcn_raw_buffer_store_b64(b, r, (int)x1, (int)so1, 0);
    }
}
template <bool RIGHT, class T, int C> __device__ __forceinline__ MVec<T, C> marchShift(const MVec<T, C>& a) {
    MVec<T, C> o;
#pragma unroll
    for (int c = 0; c < C; ++c) o.v[c] = dppShift<RIGHT>(a.v[c]);
    return o;
}

template <class T>
struct MarchK {            // kernel argument block
    int W, H;
    const T* qOld;         // rfree == 2: the solver's r_0;  rfree == 1: p_{k-2}
    const T* pOld; T* pNew; T* delta;
    const uint8_t* flags;  // bit 0: the pixel is an unknown (Op::kMasked)
    const T* coef;         // Op::kCoef values per pixel (Op::kCoef > 0)
    int iter;              // k
    int deltaMode;         // 2: this launch leaves delta alone;  1: it applies alpha_{k-2} p_{k-2} + alpha_{k-1} p_{k-1}
    int rfree;
    const T* alphaIn; T* alphaOut;      // alpha_{k-2}, beta_{k-2} as the previous launch computed them ([0], [2]) / where this launch leaves its own
    const double *aNumPrev, *aDenPrev, *s2Prev, *s3Prev; int nNum, nDen, n2, n3;
    double *aNum, *aDen, *s2, *s3;
};


template <class T, class Op, bool FLIP, int kBlk>
__global__ __launch_bounds__(kBlk) void march_pcgIter(Op op, MarchK<T> K, int rowsPerGroup, int gx) {
    constexpr int C = Op::C, kStripW = (kBlk / kWave) * kMarchSpan;
    constexpr int kCoef = Op::kCoef, kCoefN = kCoef > 0 ? kCoef : 1;
    using Vec = MVec<T, C>; using Coef = MVec<T, kCoefN>;
    __shared__ double scratch[4 * (kBlk / kWave + 1)];
    const int bx = blockIdx.x % gx, by = blockIdx.x / gx;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int x = bx * kStripW + wave * kMarchSpan + lane - 2;
    const bool xok = x >= 0 && x < K.W;
    const bool writer = xok && lane >= 2 && lane < 2 + kMarchSpan;
    const bool hasL = x >= 1, hasR = x + 1 < K.W;
    const int yb = by * rowsPerGroup, ye = min(yb + rowsPerGroup, K.H);      // sweep coordinates
    const __amdgpu_buffer_rsrc_t bQ = iw_rsrc(K.qOld), bP = iw_rsrc(K.pOld), bN = iw_rsrc(K.pNew), bD = iw_rsrc(K.delta), bF = iw_rsrc(K.flags), bC = iw_rsrc(K.coef);
    const unsigned xc = (unsigned)min(max(x, 0), K.W - 1), xv = xc * (unsigned)(C * sizeof(T)), xcv = xc * (unsigned)(kCoefN * sizeof(T));
    const bool first = K.iter == 0;
    const bool paired = K.deltaMode == 1;
    const unsigned x3 = xc * (unsigned)(3 * sizeof(T)), x1 = xc * (unsigned)sizeof(T), part1 = (unsigned)(3ull * (unsigned)K.W * (unsigned)K.H * sizeof(T));      // (kSplit31)
    auto vload = [&](__amdgpu_buffer_rsrc_t b, unsigned row) -> Vec {
        if constexpr (Op::kSplit31) return marchLoadSplit<T>(b, x3, row * (unsigned)(3 * sizeof(T)), x1, part1 + row * (unsigned)sizeof(T));
        else return marchLoad<T, C>(b, xv, row * (unsigned)(C * sizeof(T)));
    };
    auto vstore = [&](__amdgpu_buffer_rsrc_t b, unsigned row, const Vec& v) {
        if constexpr (Op::kSplit31) marchStoreSplit<T>(b, x3, row * (unsigned)(3 * sizeof(T)), x1, part1 + row * (unsigned)sizeof(T), v);
        else marchStore<T, C>(b, xv, row * (unsigned)(C * sizeof(T)), v);
    };
    struct Raw { Vec p, q, d; Coef c; int f; };      // one pixel's loads, untouched (any ALU op here would force a wait before the loop back-edge)
    auto loadRow = [&](int y) {
        Raw r;
        const int yc = min(max(y, 0), K.H - 1);      // clamped: always a valid address; rows outside the image are switched off where they enter the window
        const unsigned row = (unsigned)(FLIP ? K.H - 1 - yc : yc) * (unsigned)K.W;      // wave-uniform
        r.f = Op::kMasked ? (int)__builtin_amdgcn_raw_buffer_load_b8(bF, (int)xc, (int)row, 0) : 1;
        r.p = vload(bP, row);
        r.q = vload(bQ, row);
        if (paired) r.d = vload(bD, row); else r.d = Vec{};
        if constexpr (kCoef > 0) r.c = marchLoad<T, kCoefN>(bC, xcv, row * (unsigned)(kCoefN * sizeof(T))); else r.c = Coef{};
        return r;
    };
    // The first five rows are requested before anything else: they do not depend on the scalars of the previous launch, so their latency overlaps the prologue's
    // own memory round trip (the partial sums another kernel just wrote).
    const Raw raw0 = loadRow(yb - 2), raw1 = loadRow(yb - 1);
    Raw rwA = loadRow(yb), rwB = loadRow(yb + 1), rwC = loadRow(yb + 2);
    T alpha = 0, beta = 0;
    if (!first) {
        const double* const ps[4] = {K.aNumPrev, K.aDenPrev, K.s2Prev, K.s3Prev}; const int ns[4] = {K.nNum, K.nDen, K.n2, K.n3}; double o4[4];
        sumPartialsN<4>(ps, ns, scratch, o4);
        const double aNumD = o4[0], aDenD = o4[1], s2 = o4[2], s3 = o4[3];
        const T aNum = (T)aNumD, aDen = (T)aDenD;
        alpha = (aDen > T(0)) ? aNum / aDen : T(0);
        // The reference's start-up quirk: PCGInit1 leaves p_0 = r_0 / 4 and alphaNumerator_0 = r_0 . p_0 although later z = r (guardedInvert(1) = 1/4,
        // solverGPUGaussNewton.t:323-332, 384-392): launch 1 expands betaNumerator_0 = sum r_1^2 from 4 alphaNumerator_0 = sum r_0^2 (exact: a power of two).
        const double rr = (K.iter == 1) ? 4.0 * aNumD : aNumD;
        const double bNumD = fmax(rr - 2.0 * (double)alpha * s2 + (double)alpha * (double)alpha * s3, 0.0);      // the direct sum is >= 0: clamp cancellation noise
        beta = (aNum > T(0)) ? (T)bNumD / aNum : T(0);
    }
    if (K.alphaOut && blockIdx.x == 0 && threadIdx.x == 0) { K.alphaOut[0] = alpha; K.alphaOut[2] = beta; }
    const T alpha2 = paired ? K.alphaIn[0] : T(0);
    const bool reconR = K.rfree == 1;
    const T betaOlder = reconR ? K.alphaIn[2] : T(0);      // the beta of the previous launch: p_{k-1} = r_{k-1} + betaOlder p_{k-2}
    double accDen = 0, accNum = 0, acc2 = 0, acc3 = 0;

    struct Row { Vec p, r; Coef c; bool on; };      // iteration k-1: p_{k-1}, r_{k-1} (and the pixel's operator coefficients);  iteration k: p_k, r_k
    // raw row y enters the window: the pixel is switched off outside the image and where the mask says so; r_{k-1} is rebuilt; the row's delta gets its two terms
    auto makeOld = [&](const Raw& w, int y, Row& o) {
        o.on = xok && y >= 0 && y < K.H && (regCopy(w.f) & 1);
        Vec pv, qv;      // real copies: the raw registers are free for the next request (iw_device.h regCopy)
        if constexpr (kCoef > 0) {
#pragma unroll
            for (int c = 0; c < kCoefN; ++c) o.c.v[c] = regCopy(w.c.v[c]);
        } else o.c = Coef{};
#pragma unroll
        for (int c = 0; c < C; ++c) { pv.v[c] = regCopy(w.p.v[c]); qv.v[c] = regCopy(w.q.v[c]); }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            o.p.v[c] = o.on ? pv.v[c] : T(0);
            o.r.v[c] = o.on ? (reconR ? pv.v[c] - betaOlder * qv.v[c] : qv.v[c]) : T(0);
        }
        if (paired && writer && y >= yb && y < ye) {      // delta += alpha_{k-2} p_{k-2} + alpha_{k-1} p_{k-1} in the reference's order (solverGPUGaussNewton.t:461-462, twice)
            Vec d;
#pragma unroll
            for (int c = 0; c < C; ++c) { T t = regCopy(w.d.v[c]); t += alpha2 * qv.v[c]; t += alpha * pv.v[c]; d.v[c] = t; }
            const unsigned row = (unsigned)(FLIP ? K.H - 1 - y : y) * (unsigned)K.W;
            vstore(bD, row, d);
        }
    };
    auto applyAt = [&](const Row& c, const Row& prev, const Row& next, int y, const Coef& cf) {      // (J^T J p) of row y of a stream; prev / next in sweep order
        const Vec pl = marchShift<true>(c.p), pr = marchShift<false>(c.p);
        const bool hasPrev = y - 1 >= 0, hasNext = y + 1 < K.H;
        Vec o = FLIP ? op.apply(c.p, pl, pr, next.p, prev.p, hasL, hasR, hasNext, hasPrev, cf) : op.apply(c.p, pl, pr, prev.p, next.p, hasL, hasR, hasPrev, hasNext, cf);
#pragma unroll
        for (int i = 0; i < C; ++i) o.v[i] = c.on ? o.v[i] : T(0);
        return o;
    };
    // One trip: the freshly entered row y+2 -> A p_{k-1}(y+1), r_k, p_k (y+1) -> A p_k(y).
    // oA, oB, oC = p_{k-1} rows y, y+1, y+2;  nA, nB = p_k rows y-1, y (nC receives y+1)
    auto trip = [&](int y, const Row& oA, const Row& oB, const Row& oC, const Row& nA, const Row& nB, Row& nC, bool live) {
        const Vec ap = applyAt(oB, oA, oC, y + 1, oB.c);                               // Step1 of iteration k-1 again
        nC.on = oB.on; nC.c = Coef{};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const T r = first ? oB.r.v[c] : oB.r.v[c] - alpha * ap.v[c];                // Step2
            nC.r.v[c] = r;
            nC.p.v[c] = first ? oB.p.v[c] : r + beta * oB.p.v[c];                       // Step3 (launch 0: p_0 as PCGInit1 left it)
        }
        if (live && writer && y + 1 >= yb && y + 1 < ye) {
            const unsigned row = (unsigned)(FLIP ? K.H - 2 - y : y + 1) * (unsigned)K.W;
            vstore(bN, row, nC.p);
        }
        const Vec o = applyAt(nB, nA, nC, y, oA.c);                                     // Step1 of iteration k (row y: the coefficients of oA's pixel)
        if (live && writer && y >= yb) {
            // every term from the same r, A p in double, where a product of two floats is exact: the expansion of the beta numerator cancels to as many digits as
            // the residual loses in one iteration
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double r = (double)nB.r.v[c], a = (double)o.v[c], p = (double)nB.p.v[c];
                accNum += (first ? p : r) * r;      // z_0 . r_0 is the reference's r_0 . p_0
                accDen += p * a; acc2 += r * a; acc3 += a * a;
            }
        }
    };
    Row o0, o1, o2, n0{}, n1{}, n2{};
    makeOld(raw0, yb - 2, o0);
    makeOld(raw1, yb - 1, o1);
    // trips y = yb-2 .. ye-1 (the first two only build p_k(yb-1), p_k(yb)); three per pass, no branch around a load; the barrier keeps the strip's waves on the same rows
    for (int y = yb - 2; y < ye; y += 3) {
        __syncthreads();
        { makeOld(rwA, y + 2, o2); rwA = loadRow(y + 5); trip(y, o0, o1, o2, n0, n1, n2, true); }
        { makeOld(rwB, y + 3, o0); rwB = loadRow(y + 6); trip(y + 1, o1, o2, o0, n1, n2, n0, y + 1 < ye); }
        { makeOld(rwC, y + 4, o1); rwC = loadRow(y + 7); trip(y + 2, o2, o0, o1, n2, n0, n1, y + 2 < ye); }
    }
    double v[4] = {accDen, accNum, acc2, acc3};
    blockReduceSumN<4>(v, scratch);
    if (threadIdx.x == 0) { K.aDen[blockIdx.x] = v[0]; K.aNum[blockIdx.x] = v[1]; K.s2[blockIdx.x] = v[2]; K.s3[blockIdx.x] = v[3]; }
}

// delta += alpha p for the term an odd last launch still owes (alpha as that launch left it)
template <class T>
__global__ __launch_bounds__(kBlock) void march_axpyDeferred(T* __restrict__ delta, const T* __restrict__ p, const T* __restrict__ alpha, long n) {
    const T a = alpha[0];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) delta[i] += a * p[i];
}

// Host side of the loop: the ring of three p buffers, the alpha / beta slots, sweep direction and the deferred delta term (the bookkeeping of
// ImageWarpingOps::pcgIteration, energy_image_warping.hip).
template <class T>
struct MarchLoop {
    T* ring[3] = {nullptr, nullptr, nullptr}; const T* r0Ptr = nullptr; T* alphaSlots = nullptr;
    int iterIndex = 0, flip = 0, occ = 0, forceRows = 0, forceBlock = 0; bool deferredTerm = false;
    MarchLoop() { if (const char* e = getenv("OPT_AMD_ITER_ROWS")) forceRows = atoi(e); forceBlock = devSwitch("OPT_AMD_MARCH_BLOCK", forceBlock); }
    ~MarchLoop() { for (T* b : ring) if (b) (void)hipFree(b); if (alphaSlots) (void)hipFree(alphaSlots); }
    template <class Op>
    bool launch(const Op& op, int W, int H, const uint8_t* flags, int cus, const PcgIterArgs<T>& a, LaunchCtx& ctx, const T* coef = nullptr) {
        // 16-byte pixels leave room for 3 waves per SIMD in 768-thread workgroups (12 column strips side by side: fewer, fatter workgroups and a quarter of the partial
        // sums the next prologue has to add): best at 2048^2 (poisson 55.8 us against 58.6 with 512 threads, 60.8 with 256); narrower images do better with 512
        // (1024^2: optical_flow 21.4 / 21.8 / 25.2 us for 512 / 768 / 256, intrinsic 28.2 / 32.5 / 30.9; poisson 256^2 11.9 / 13.6 / 13.0); 32-byte pixels
        // (double4: 206 VGPRs) run 256 threads
        const int blk = forceBlock ? forceBlock : (Op::C * sizeof(T) <= 16 ? (W >= 1536 ? 768 : 512) : 256);
        if constexpr (Op::kMaxBlock >= 768) { if (blk == 768) return launchB<Op, 768>(op, W, H, flags, cus, a, ctx, coef); }      // (Op::kMaxBlock: wider workgroups would spill and are not instantiated)
        if constexpr (Op::kMaxBlock >= 512) { if (blk >= 512) return launchB<Op, 512>(op, W, H, flags, cus, a, ctx, coef); }
        return launchB<Op, 256>(op, W, H, flags, cus, a, ctx, coef);
    }
    template <class Op, int blk>
    bool launchB(const Op& op, int W, int H, const uint8_t* flags, int cus, const PcgIterArgs<T>& a, LaunchCtx& ctx, const T* coef) {
        constexpr int C = Op::C;
        if ((unsigned long long)W * H * C * sizeof(T) >= (1ull << 32)) return false;      // 32-bit buffer offsets
        const size_t bytes = ((size_t)W * H * C + 3) / 4 * 4 * sizeof(T);                 // padded like the solver's vectors: its flat kernels read whole 16-byte packs of the last p
        for (int j = 0; j < 3; ++j) if (!ring[j]) { HIP_CHECK(hipMalloc((void**)&ring[j], bytes)); HIP_CHECK(hipMemsetAsync(ring[j], 0, bytes, ctx.stream)); }
        if (!alphaSlots) { HIP_CHECK(hipMalloc((void**)&alphaSlots, 4 * sizeof(T))); HIP_CHECK(hipMemsetAsync(alphaSlots, 0, 4 * sizeof(T), ctx.stream)); }   // [0,1] alpha, [2,3] beta, ping-pong
        if (a.first) { iterIndex = 0; flip = 0; r0Ptr = a.rOld; }      // the solver swaps its r buffers after every launch; this one keeps r_0 until launch 1 has read it
        if (occ == 0) {
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, march_pcgIter<T, Op, false, blk>, blk, 0));
            occ = std::max(1, std::min(occ, 8));
        }
        const int k = iterIndex;
        MarchK<T> K{};
        K.W = W; K.H = H; K.flags = flags; K.coef = coef; K.iter = k;
        K.pOld = k == 0 ? a.pOld : ring[(k - 1) % 3];
        K.qOld = k <= 1 ? r0Ptr : ring[(k - 2) % 3];
        K.pNew = ring[k % 3]; K.delta = a.delta;
        K.rfree = k <= 1 ? 2 : 1;
        K.deltaMode = (k >= 2 && k % 2 == 0) ? 1 : 2;           // launch 0 has nothing to apply; odd launches defer
        K.alphaOut = alphaSlots + (k & 1); K.alphaIn = alphaSlots + ((k & 1) ^ 1);
        deferredTerm = k >= 1 && k % 2 == 1;                    // after an odd launch alpha_{k-1} p_{k-1} is still owed (finish)
        K.aNumPrev = a.aNumPrev.partials; K.aDenPrev = a.aDenPrev.partials; K.s2Prev = a.s2Prev.partials; K.s3Prev = a.s3Prev.partials;
        K.nNum = a.aNumPrev.n; K.nDen = a.aDenPrev.n; K.n2 = a.s2Prev.n; K.n3 = a.s3Prev.n;
        K.aNum = a.aNum->partials; K.aDen = a.aDen->partials; K.s2 = a.s2->partials; K.s3 = a.s3->partials;
        const int gx = divUp(W, (blk / kWave) * kMarchSpan);
        const int target = std::max(gx, cus * occ);
        int gy = std::max(1, std::min(std::min(H, target / gx), kMaxPartials / gx));
        int rowsPerGroup = divUp(H, gy);
        if (forceRows > 0) rowsPerGroup = std::max(divUp(H, std::max(1, kMaxPartials / gx)), std::min(H, forceRows));
        gy = divUp(H, rowsPerGroup);
        {
            ScopedKernel sk(ctx, "PCGIteration");
            if (flip) march_pcgIter<T, Op, true, blk><<<gx * gy, blk, 0, ctx.stream>>>(op, K, rowsPerGroup, gx);
            else march_pcgIter<T, Op, false, blk><<<gx * gy, blk, 0, ctx.stream>>>(op, K, rowsPerGroup, gx);
        }
        flip ^= 1;      // successive launches sweep top-down / bottom-up
        ++iterIndex;
        a.aNum->n = a.aDen->n = a.s2->n = a.s3->n = gx * gy;
        return true;
    }
    // After the last launch L-1 of a linear solve: the deferred term alpha_{L-2} p_{L-2} of an odd last launch; returns where p_{L-1} lives (the solver adds alpha_{L-1} p_{L-1})
    const T* finish(T* delta, long n, int cus, LaunchCtx& ctx) {
        if (iterIndex < 1) return nullptr;
        const T* pLast = ring[(iterIndex - 1) % 3];
        if (deferredTerm && iterIndex >= 2) {
            ScopedKernel sk(ctx, "PCGStep2_delta");
            const int g = (int)std::max<long>(1, std::min<long>((n + kBlock - 1) / kBlock, (long)cus * 8));
            march_axpyDeferred<T><<<g, kBlock, 0, ctx.stream>>>(delta, ring[(iterIndex - 2) % 3], alphaSlots + ((iterIndex - 1) & 1), n);
        }
        deferredTerm = false;
        return pLast;
    }
};

}  // namespace
}  // namespace optamd
