// 5-point stencils with C channels per pixel (the operators of stencil_march.h: poisson_image_editing, the minimal laplacian, optical_flow, intrinsic_image_decomposition): the WHOLE PCG linear solve
// of a Gauss-Newton step as one persistent launch whose loop state never leaves the chip.
//
// What it replaces: the reference's loop `for lIter = 0, lIterations do PCGStep1; PCGStep2; PCGStep3 end` (solverGPUGaussNewton.t:1056-1092) -- one marching launch per
// iteration in march_pcgIter, 11-16 us each on images that are all launch latency (BASELINE config 1: poisson 256^2).  The protocol is sfs_onchip.h's with a one-pixel ring:
//   tile    a WAVE holds 64 columns x (R + 2) rows of p and r in registers and owns the 62 x R pixels in the middle; the ring is updated by the holder with the owner's
//           alpha, beta and the same fused operations, so the search direction never travels;
//   A p     Op::apply on the owned rows (neighbouring columns: whole-wave DPP shifts; rows above / below: the lane's own registers); flag bit and operator coefficients
//           of the held pixels stay in registers for the whole solve;
//   ring    the A p of a tile's outermost rows / columns goes to a tagged image (8-byte {payload, tag} words, relaxed agent-scope stores, parity-double-buffered) and is
//           picked up by the ring holders inside the ONE wait per iteration that also carries the four sums (alphaNum, alphaDen, s2, s3; beta by expansion as in
//           march_pcgIter, including the reference's start: p_0 = r_0 / 4, alphaNumerator_0 = r_0 . p_0, so sum r_0^2 = 4 alphaNumerator_0 exactly);
//   sums    every workgroup posts its partial sums as tagged words and adds ALL workgroups' words in the same order: the same bits everywhere.
// Every wait is bounded by the device's wall clock; a time-out raises `bad`, nothing is written to delta, the unknowns stay untouched (march_applyDelta checks the flag) and
// the host redoes the linear solve with the marching kernels.  The grid must be co-resident (one workgroup per CU): the launcher checks workgroups <= CUs.
// Levenberg-Marquardt (LM = true): + CtC p (o.t:2076-2082; CtC as PCGFinalizeDiagonal left it, the start p_0 = M_LM r_0 comes from the solver, later z = r), a fifth sum --
// sum r_0^2 in iteration 0, then Q_k = 1/2 sum delta . (r + b) (solver.t:483-485) formed where iteration k is applied and carried by the sums of iteration k + 1 -- and the
// q early-out (:1093-1102) decided by every workgroup from the same totals; a residual reset before the last iteration (lIterations > residual_reset_period) keeps the solve
// on the generic kernels.  Op::kSplit31 (intrinsic_image_decomposition: two unknown images) only changes where a pixel's scalars sit in the solver's vectors.
#pragma once
#include "stencil_march.h"
#include "onchip_sync.h"

namespace optamd {
namespace {

constexpr int kMoSpan = kWave - 2;            // pixels a wave owns per row (one DPP ring)
constexpr int kMoMaxG = 256;                  // workgroups (one per CU)
constexpr int kMoNSMax = 5, kMoNWMax = 2 * kMoNSMax;   // sums per iteration (Gauss-Newton 4, Levenberg-Marquardt 5); tagged words per workgroup

template <class T>
struct MoArgs {
    int W, H;
    const T* r0; const T* p0; T* delta;     // solver vectors: C channels per pixel, interleaved
    const uint8_t* flags; const T* coef;    // Op::kMasked / Op::kCoef
    int stripsX, tilesY, G, L;
    unsigned tag0;                          // tag of iteration 0 (tags never repeat over the life of the buffers)
    oc_u64* slots;                          // [2][G][8]
    oc_u64* apBox;                          // [2][W * H * C * sizeof(T) / 4]
    int* bad; long long timeoutTicks; int failAt;
    long long firstTicks;      // bound of the FIRST iteration's wait: the co-residency check (every workgroup has posted its words once it passes), before anything is written
    const T* CtC; T qTolerance; int* hostErr;      // LM: the clamped diagonal, q_tolerance, the pinned word a workgroup that gave up raises (the solver applies the update itself)
    double* lmBreak;                               // pinned {iteration + 1, zeta} of the q early-out (OnChipLm::breakInfo), or nullptr
};

__device__ __forceinline__ float moFma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double moFma(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <class T, class Op, int R, int WAVES, bool LM>
__global__ __launch_bounds__(WAVES * kWave) void march_onchipPcg(Op op, MoArgs<T> K) {
    constexpr int kMoNS = LM ? 5 : 4, kMoNW = 2 * kMoNS;
    constexpr int C = Op::C, HR = R + 2, kBlk = WAVES * kWave, WPS = (int)sizeof(T) / 4;
    constexpr int kCoefN = Op::kCoef > 0 ? Op::kCoef : 1;
    using Vec = MVec<T, C>; using Coef = MVec<T, kCoefN>;
    __shared__ double red[kMoNS * WAVES];
    __shared__ double TOT[kMoNS + 1];
    __shared__ unsigned W1[kMoMaxG * kMoNW];
    constexpr bool AP_LDS = C * sizeof(T) * R >= 128;      // where the registers are short, A p of the owned pixels waits in LDS between the stencil and the update: [row][channel][thread]
    __shared__ T apL[AP_LDS ? R * C * kBlk : 1];
    constexpr bool DL_LDS = AP_LDS && Op::kCoef >= 4 && (C + Op::kCoef) * sizeof(T) * R >= 256;      // ... and, for the fattest pixels (four channels + four coefficients), delta itself
    __shared__ T dlL[DL_LDS ? R * C * kBlk : 1];
    __shared__ T bL[LM ? R * C * kBlk : 1];      // LM: b = r_0 of the owned pixels (for Q)
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = blockIdx.x;
    const int tile = g * WAVES + wave;
    const int sx = tile % K.stripsX, ty = tile / K.stripsX;
    const bool idle = ty >= K.tilesY;                  // (wave-uniform) a wave without a tile: contributes zeros to the sums
    const int x = sx * kMoSpan + lane - 1;
    const int yBase = ty * R;                          // first owned row; held row h is image row yBase - 1 + h
    const bool xin = !idle && x >= 0 && x < K.W;
    const bool writer = xin && lane >= 1 && lane <= kMoSpan;
    const bool hasL = x >= 1, hasR = x + 1 < K.W;
    const int xc = min(max(x, 0), K.W - 1);
    const int N = K.W * K.H;
    int* const bad = K.bad;
    const long long to = K.timeoutTicks;
    // scalar (pixel i, channel c) of a solver vector: C interleaved channels, or -- Op::kSplit31 -- a 3-channel image followed by a 1-channel image (energy.h)
    auto at = [&](long i, int c) -> long { if constexpr (Op::kSplit31) return c < 3 ? i * 3 + c : 3L * N + i; else return i * C + c; };

    // ---- p_0, r_0, the flag bit and the operator coefficients of the held pixels (a pixel outside the image or switched off: zeros, off); delta = 0 ----------------
    Vec p[HR], r[HR], dl[DL_LDS ? 1 : R], ap[AP_LDS ? 1 : R];
    Coef cf[HR];
    Vec ctc[LM ? R : 1];      // LM: CtC of the owned pixels
    unsigned onBits = 0;
#pragma unroll
    for (int h = 0; h < HR; ++h) {
        const int y = yBase - 1 + h;
        const bool in = xin && y >= 0 && y < K.H;
        const long i = in ? (long)y * K.W + xc : (long)xc;      // a valid address either way
        bool on = in;
        if (Op::kMasked) on = on && (K.flags[i] & 1);
#pragma unroll
        for (int c = 0; c < C; ++c) { const T pv = K.p0[at(i, c)], rv = K.r0[at(i, c)]; p[h].v[c] = on ? pv : T(0); r[h].v[c] = on ? rv : T(0); }
#pragma unroll
        for (int c = 0; c < kCoefN; ++c) cf[h].v[c] = Op::kCoef > 0 ? K.coef[i * kCoefN + c] : T(0);
        onBits |= on ? (1u << h) : 0u;
        if (LM && h >= 1 && h <= R) {
#pragma unroll
            for (int c = 0; c < C; ++c) { const T cv = K.CtC[at(i, c)]; ctc[LM ? h - 1 : 0].v[c] = on ? cv : T(0); bL[((LM ? h - 1 : 0) * C + c) * kBlk + tid] = r[h].v[c]; }      // b = r_0 (solver.t:657)
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int c = 0; c < C; ++c) { if (DL_LDS) dlL[((DL_LDS ? i : 0) * C + c) * kBlk + tid] = 0; else dl[DL_LDS ? 0 : i].v[c] = 0; if (AP_LDS) apL[((AP_LDS ? i : 0) * C + c) * kBlk + tid] = 0; else ap[AP_LDS ? 0 : i].v[c] = 0; }

    const int pixBase = (yBase - 1) * K.W + xc;      // index of held row 0 of this lane's column (used only where the row exists)
    auto rowIn = [&](int h) { const int y = yBase - 1 + h; return xin && y >= 0 && y < K.H; };
    bool failed = false;
    double accQ = 0;
    T Q0 = 0;      // fetchQ before the loop (solver.t:1050): delta = 0, so exactly 0
    const size_t boxStride = (size_t)N * C * WPS;

    for (int k = 0; k < K.L; ++k) {
        const unsigned tag = K.tag0 + (unsigned)k;
        const int par = (int)(tag & 1u);
        oc_u64* const box = K.apBox + (size_t)par * boxStride;
        oc_u64* const slotPar = K.slots + (size_t)par * K.G * kMoNW;
        if (k == K.failAt && g == 0 && tid == 0) __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool first = k == 0;

        // ---- PCGStep1: A p_k on the owned pixels, with the four sums (march_pcgIter's expressions) --------------------------------------------------------------
        double accDen = 0, accNum = 0, acc2 = 0, acc3 = 0, accX = 0;      // accX (LM): sum r_0^2 in iteration 0, the Q of the iteration before in the others
        if (!idle) {
#pragma unroll
            for (int h = 1; h <= R; ++h) {
                const int y = yBase - 1 + h;
                const Vec pl = marchShift<true>(p[h]), pr = marchShift<false>(p[h]);
                Vec o = op.apply(p[h], pl, pr, p[h - 1], p[h + 1], hasL, hasR, y - 1 >= 0, y + 1 < K.H, cf[h]);
                const bool on = (onBits >> h) & 1u;
#pragma unroll
                for (int c = 0; c < C; ++c) { if (LM) o.v[c] += ctc[LM ? h - 1 : 0].v[c] * p[h].v[c]; o.v[c] = on ? o.v[c] : T(0); if (AP_LDS) apL[((AP_LDS ? h - 1 : 0) * C + c) * kBlk + tid] = o.v[c]; else ap[AP_LDS ? 0 : h - 1].v[c] = o.v[c]; }
                if (writer && y < K.H) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const double rr = (double)r[h].v[c], a = (double)o.v[c], pp = (double)p[h].v[c];
                        accNum += (first ? pp : rr) * rr;      // z_0 . r_0 is the reference's r_0 . p_0
                        accDen += pp * a; acc2 += rr * a; acc3 += a * a;
                        if (LM && first) accX += rr * rr;
                    }
                    // the tile's outermost rows / columns: to the tagged image, for whoever holds them as ring
                    if (h == 1 || h == R || lane == 1 || lane == kMoSpan) {
                        const size_t i = (size_t)(pixBase + h * K.W) * C * WPS;
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            if constexpr (WPS == 1) ocStore(box + i + c, tag, __float_as_uint((float)o.v[c]));
                            else { const oc_u64 b = (oc_u64)__double_as_longlong((double)o.v[c]); ocStore(box + i + 2 * c, tag, (unsigned)b); ocStore(box + i + 2 * c + 1, tag, (unsigned)(b >> 32)); }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- the grid-wide sums; the ring's A p is collected inside the wait ------------------------------------------------------------------------------------
        {
            if (LM && !first) accX = accQ;
            double v4[kMoNS];
            v4[0] = accNum; v4[1] = accDen; v4[2] = acc2; v4[3] = acc3;
            if constexpr (LM) v4[4] = accX;
#pragma unroll
            for (int q = 0; q < kMoNS; ++q) { v4[q] = ocWaveSum63(v4[q]); if (lane == kWave - 1) red[q * WAVES + wave] = v4[q]; }
        }
        __syncthreads();
        if (tid < kMoNW) {
            double s = 0;
            for (int w = 0; w < WAVES; ++w) s += red[(tid >> 1) * WAVES + w];
            const oc_u64 b = (oc_u64)__double_as_longlong(s);
            ocStore(slotPar + (size_t)g * kMoNW + tid, tag, (tid & 1) ? (unsigned)(b >> 32) : (unsigned)b);
        }
        Vec ring[HR];
        {
            constexpr int kPer = (kMoMaxG * kMoNW + kBlk - 1) / kBlk;
            oc_u64 w[kPer];
            const int nW = K.G * kMoNW;
            const bool lastIt = k + 1 == K.L;      // (after the last iteration only delta survives: nobody needs the ring)
            // Ring requests: every lane asks for its column's pixel of the rows above and below the tile; the two side columns are asked for by ONE lane per pixel
            // (lane h: the left neighbour of row h, lane 32 + h: the right one) and handed to lanes 0 / 63 through scalar registers afterwards -- a lane holds three
            // pixels' words during the wait instead of R + 2 (the difference is what lets 16 rows per wave fit).
            static_assert(R + 1 < 32, "one lane per side pixel");
            const int sRow = lane & 31, sX = (lane < 32) ? sx * kMoSpan - 1 : sx * kMoSpan + kMoSpan, sY = yBase - 1 + sRow;
            const bool needTop = !lastIt && rowIn(0), needBot = !lastIt && rowIn(HR - 1);
            const bool needSide = !lastIt && !idle && sRow >= 1 && sRow <= R && sX >= 0 && sX < K.W && sY < K.H;
            const size_t iTop = (size_t)pixBase * C * WPS, iBot = (size_t)(pixBase + (HR - 1) * K.W) * C * WPS, iSide = needSide ? ((size_t)sY * K.W + sX) * C * WPS : 0;
            oc_u64 rw[3][C * WPS];      // top, bottom, side
            bool sumsOk = false, ringOk = false;
            auto askSums = [&]() {
#pragma unroll
                for (int u = 0; u < kPer; ++u) { const int i = tid + u * kBlk; w[u] = ocLoad(slotPar + (i < nW ? i : tid % nW)); }
            };
            auto askRing = [&]() {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const bool nd = j == 0 ? needTop : j == 1 ? needBot : needSide;
                    const size_t i = j == 0 ? iTop : j == 1 ? iBot : iSide;
#pragma unroll
                    for (int q = 0; q < C * WPS; ++q) rw[j][q] = (oc_u64)tag << 32;
                    if (nd) {
#pragma unroll
                        for (int q = 0; q < C * WPS; ++q) rw[j][q] = ocLoad(box + i + q);
                    }
                }
            };
            auto check = [&]() {
                if (!sumsOk) {
                    bool ok = true;
#pragma unroll
                    for (int u = 0; u < kPer; ++u) { const int i = tid + u * kBlk; ok = ok && (i >= nW || (unsigned)(w[u] >> 32) == tag); }
                    sumsOk = ok;
                }
                if (!ringOk) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int q = 0; q < C * WPS; ++q) ok = ok && (unsigned)(rw[j][q] >> 32) == tag;
                    ringOk = ok;
                }
                return sumsOk && ringOk;
            };
            askSums(); askRing();
            if (!check()) {
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                for (;;) {
                    __builtin_amdgcn_s_sleep(1);
                    if (!sumsOk) askSums();
                    if (!ringOk) askRing();
                    if (check()) break;
                    if ((++spins & 31u) == 0) {
                        if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        if (wall_clock64() - t0 > (k == 0 ? K.firstTicks : to)) { __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                }
            }
            auto decode = [&](const oc_u64 (&q)[C * WPS], int c) -> T {
                if constexpr (WPS == 1) return __uint_as_float((unsigned)q[c]);
                else return __longlong_as_double((long long)((q[2 * c + 1] << 32) | (q[2 * c] & 0xffffffffull)));
            };
            auto laneOf = [](T v, int l) -> T {
                if constexpr (sizeof(T) == 8) return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
                else return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
            };
#pragma unroll
            for (int c = 0; c < C; ++c) {
                ring[0].v[c] = decode(rw[0], c); ring[HR - 1].v[c] = decode(rw[1], c);
                const T sv = decode(rw[2], c);
#pragma unroll
                for (int h = 1; h <= R; ++h) { const T lft = laneOf(sv, h), rgt = laneOf(sv, 32 + h); ring[h].v[c] = lane == 0 ? lft : rgt; }      // (only lanes 0 and 63 use them)
            }
#pragma unroll
            for (int u = 0; u < kPer; ++u) { const int i = tid + u * kBlk; if (i < nW) W1[i] = (unsigned)w[u]; }
            __syncthreads();
            // every workgroup adds all workgroups' words in the same order: wave q takes sum q, a lane the workgroups lane, lane + 64, lane + 128, lane + 192 in that
            // order, then the wave's DPP tree -- the same association everywhere, so the same bits
#pragma unroll
            for (int pass = 0; pass < (kMoNS + WAVES - 1) / WAVES; ++pass) {      // (five sums on four waves: wave 0 takes the fifth as well)
                const int q = wave + pass * WAVES;
                if (q < kMoNS) {
                    double sacc = 0;
#pragma unroll
                    for (int c = 0; c < kMoMaxG / kWave; ++c) {
                        const int m = lane + c * kWave;
                        const double v = m < K.G ? ocJoin(W1[m * kMoNW + 2 * q], W1[m * kMoNW + 2 * q + 1]) : 0.0;
                        sacc += v;
                    }
                    sacc = ocWaveSum63(sacc);
                    if (lane == kWave - 1) TOT[q] = sacc;
                }
            }
            if (tid == 0) reinterpret_cast<int*>(TOT + kMoNS)[0] = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
        const double aNumD = TOT[0], aDenD = TOT[1], s2 = TOT[2], s3 = TOT[3];
        if (reinterpret_cast<const int*>(TOT + kMoNS)[0]) { failed = true; break; }      // uniform over the workgroup: a wait timed out somewhere
        if constexpr (LM) {      // the q early-out of iteration k - 1 (solver.t:1093-1102): nothing of iteration k has been applied yet
            if (!first) {
                const T Q1 = (T)TOT[4];
                const T zeta = T(k) * (Q1 - Q0) / Q1;
                if (zeta < K.qTolerance) { if (K.lmBreak && blockIdx.x == 0 && tid == 0) { K.lmBreak[1] = (double)zeta; K.lmBreak[0] = (double)(k + 1); } break; }
                Q0 = Q1;
            }
        }
        // the scalars of march_pcgIter's prologue (solver.t:456-459, 544-547 guards; beta numerator by expansion, clamped like the direct sum it replaces; the start-up
        // quirk: sum r_0^2 = 4 alphaNumerator_0, exact)
        const T aNum = (T)aNumD, aDen = (T)aDenD;
        const T alpha = (aDen > T(0)) ? aNum / aDen : T(0);
        const double rr = first ? (LM ? TOT[LM ? 4 : 0] : 4.0 * aNumD) : aNumD;      // (LM starts from the preconditioned r_0: sum r_0^2 is summed directly)
        const double bNumD = fmax(rr - 2.0 * (double)alpha * s2 + (double)alpha * (double)alpha * s3, 0.0);
        const T beta = (aNum > T(0)) ? (T)bNumD / aNum : T(0);
        const bool last = k + 1 == K.L;

        // ---- PCGStep2 + PCGStep3 (z = r): delta += alpha p;  r -= alpha A p;  p = r + beta p -- on the owned pixels and, with the same fused operations, on the ring
        accQ = 0;
#pragma unroll
        for (int h = 0; h < HR; ++h) {
            const bool ownRow = h >= 1 && h <= R;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const T apv = ownRow ? (writer ? (AP_LDS ? apL[((AP_LDS && ownRow ? h - 1 : 0) * C + c) * kBlk + tid] : ap[!AP_LDS && ownRow ? h - 1 : 0].v[c]) : ring[h].v[c]) : ring[h].v[c];
                T dNew = 0;
                if (ownRow) {
                    dNew = moFma(alpha, p[h].v[c], DL_LDS ? dlL[((DL_LDS && ownRow ? h - 1 : 0) * C + c) * kBlk + tid] : dl[!DL_LDS && ownRow ? h - 1 : 0].v[c]);
                    if (DL_LDS) dlL[((DL_LDS && ownRow ? h - 1 : 0) * C + c) * kBlk + tid] = dNew; else dl[!DL_LDS && ownRow ? h - 1 : 0].v[c] = dNew;
                }
                if (!last) {
                    r[h].v[c] = moFma(-alpha, apv, r[h].v[c]);
                    if (LM && ownRow && writer && yBase - 1 + h < K.H) accQ += (double)(T(0.5) * (dNew * (r[h].v[c] + bL[((LM && ownRow ? h - 1 : 0) * C + c) * kBlk + tid])));      // solver.t:483-485
                    p[h].v[c] = moFma(beta, p[h].v[c], r[h].v[c]);
                }
            }
        }
    }
    if (failed && tid == 0 && K.hostErr) __hip_atomic_store(K.hostErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!failed && writer) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int y = yBase + i;
            if (y < K.H) {
                const long e = (long)y * K.W + x;
#pragma unroll
                for (int c = 0; c < C; ++c) K.delta[at(e, c)] = DL_LDS ? dlL[((DL_LDS ? i : 0) * C + c) * kBlk + tid] : dl[DL_LDS ? 0 : i].v[c];
            }
        }
    }
}

// PCGLinearUpdate X += delta (solver.t:552-557) behind the on-chip solve -- unless a wait timed out: then the unknowns stay untouched and the host is told
template <class T>
__global__ __launch_bounds__(kBlock) void march_applyDelta(T* __restrict__ X, const T* __restrict__ delta, long n, const int* __restrict__ bad, int* hostErr) {
    if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(hostErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) X[i] = X[i] + delta[i];
}

// Host side: buffers, variant choice, launch, the time-out verdict.  OPT_AMD_ONCHIP=0 switches the path off (the one A/B switch); OPT_AMD_ONCHIP_ROWS / _WAVES force a
// variant (tests run every variant on small images); OPT_AMD_ONCHIP_FAIL_AT / _TIMEOUT_MS: the time-out path's test hooks.
template <class T>
struct OnchipMarch {
    bool enabled = true, failed = false, launched = false;
    int forceRows = 0, forceWaves = 0, failAt = -1; long long timeoutTicks = 0;      // 0: onchip_sync.h ocTimeouts() decides; OPT_AMD_ONCHIP_TIMEOUT_MS overrides
    oc_u64 *slots = nullptr, *box = nullptr; int *bad = nullptr, *hostErr = nullptr; unsigned seq = 0; size_t slotBytes = 0, boxBytes = 0;
    int lastRows = 0, lastWaves = 0, lastG = 0;
    OnchipMarch() {
        if (const char* e = getenv("OPT_AMD_ONCHIP")) enabled = atoi(e) != 0;
        if (const char* e = getenv("OPT_AMD_ONCHIP_ROWS")) forceRows = std::max(0, atoi(e));
        if (const char* e = getenv("OPT_AMD_ONCHIP_WAVES")) forceWaves = std::max(0, atoi(e));
        if (const char* e = getenv("OPT_AMD_ONCHIP_FAIL_AT")) failAt = atoi(e);
        if (const char* e = getenv("OPT_AMD_ONCHIP_TIMEOUT_MS")) timeoutTicks = std::max(1, atoi(e)) * 100000LL;
    }
    ~OnchipMarch() { if (slots) (void)hipFree(slots); if (box) (void)hipFree(box); if (bad) (void)hipFree(bad); if (hostErr) (void)hipHostFree(hostErr); }
    // The buffers of the path, sized for the plan's image (the dimensions of a plan are fixed).  Called when the plan is made (the kernel set's constructor), so that the
    // first linear solve of a plan does not pay for four allocations; solve() calls it itself if nobody has.  zero = no tag.
    // ... only for plans that can take the path at all: some variant of THIS operator fits THIS device's CUs for the image (a 4-channel double image of
    // 1-2 M pixels used to allocate ~256 MB of tagged box it could never use)
    template <class Op> void reserveFor(int W, int H, int cus) {
        int sx, ty, G;
        if (select<Op, false>(W, H, cus, sx, ty, G) || select<Op, true>(W, H, cus, sx, ty, G)) reserve(W, H, Op::C);
    }
    void reserve(int W, int H, int C) {
        if (slots || !enabled || (unsigned long long)W * H * C * sizeof(T) >= (1ull << 30)) return;
        if ((long)W * H > (long)kMoMaxG * 8 * kMoSpan * 16 || divUp(W, kMoSpan) > kMoMaxG * 8) return;      // (more pixels than the largest variant holds on the largest grid: the path will never be taken)
        slotBytes = sizeof(oc_u64) * 2 * (size_t)kMoMaxG * kMoNWMax; boxBytes = sizeof(oc_u64) * 2 * (size_t)W * H * C * (sizeof(T) / 4);
        HIP_CHECK(hipMalloc((void**)&slots, slotBytes)); HIP_CHECK(hipMalloc((void**)&box, boxBytes));
        HIP_CHECK(hipMalloc((void**)&bad, sizeof(int))); HIP_CHECK(hipHostMalloc((void**)&hostErr, 64)); *hostErr = 0;
        HIP_CHECK(hipMemset(bad, 0, sizeof(int))); HIP_CHECK(hipMemset(slots, 0, slotBytes)); HIP_CHECK(hipMemset(box, 0, boxBytes)); HIP_CHECK(hipStreamSynchronize(nullptr));      // (done before the plan's own stream sees the buffers)
        seq = 2;
    }
    struct Variant { int rows, waves; const void* fn; };
    template <class Op, int R, int WV, bool LM> static constexpr size_t ldsBytes() {      // A p and delta (where they wait in LDS) + b (LM) + the sums' staging
        const size_t plane = (size_t)R * Op::C * sizeof(T) * WV * kWave;
        const bool apLds = Op::C * sizeof(T) * R >= 128, dlLds = apLds && Op::kCoef >= 4 && (Op::C + Op::kCoef) * sizeof(T) * R >= 256;
        return (apLds ? plane : 0) + (dlLds ? plane : 0) + (LM ? plane : 0) + 12 * 1024;
    }
    template <class Op, bool LM> static const std::vector<Variant>& variants() {
        static const std::vector<Variant> v = [] {
            std::vector<Variant> o;
#define MO_VARIANT(R, WV) if constexpr (ldsBytes<Op, R, WV, LM>() <= 150 * 1024 && !Op::template spills<R, WV, LM>()) o.push_back({R, WV, (const void*)march_onchipPcg<T, Op, R, WV, LM>})
            MO_VARIANT(2, 4); MO_VARIANT(4, 4); MO_VARIANT(8, 4); MO_VARIANT(2, 8); MO_VARIANT(4, 8); MO_VARIANT(8, 8);
            if constexpr (Op::C * sizeof(T) <= 8) { MO_VARIANT(16, 4); MO_VARIANT(16, 8); }
#undef MO_VARIANT
            // a variant whose registers do not hold its loop state is not instantiated (Op::spills, from the compiler's resource remarks); should a compiler upgrade make another one
            // spill it is still not offered: no scratch in a kernel that is all latency
            std::vector<Variant> ok;
            for (const auto& v : o) { hipFuncAttributes fa{}; if (hipFuncGetAttributes(&fa, v.fn) == hipSuccess && fa.localSizeBytes == 0) ok.push_back(v); else (void)hipGetLastError(); }
            return ok;
        }();
        return v;
    }
    // among the variants whose workgroups fit one per CU: the least marching time per SIMD and iteration
    template <class Op, bool LM = false> const Variant* select(int W, int H, int cus, int& stripsX, int& tilesY, int& G) const {
        stripsX = divUp(W, kMoSpan);
        const Variant* best = nullptr; int bestCost = 1 << 30;
        for (const auto& v : variants<Op, LM>()) {
            if (forceRows && v.rows != forceRows) continue;
            if (forceWaves && v.waves != forceWaves) continue;
            const int ty = divUp(H, v.rows), g = divUp(stripsX * ty, v.waves);
            if (g > std::min(cus, kMoMaxG)) continue;
            const int cost = (v.waves == 4 ? 100 : 136) * (v.rows + 2);      // (measured: two waves per SIMD march a row pair in 1.36 of the time one wave marches a row)
            if (cost < bestCost) { best = &v; bestCost = cost; tilesY = ty; G = g; }
        }
        return best;
    }
    // the whole linear solve + X += delta; false (nothing touched): not offered for this plan
    // lm: Levenberg-Marquardt (the solver applies the update itself and hears of a time-out through hostErr)
    template <class Op> bool solve(const Op& op, int W, int H, const uint8_t* flags, const T* coef, const T* r0, const T* p0, T* delta, T* X, int L, int cus, LaunchCtx& ctx,
                                   const OnChipLm<T>* lm = nullptr) {
        constexpr int C = Op::C;
        if (!enabled || failed || L <= 0 || (unsigned long long)W * H * C * sizeof(T) >= (1ull << 30)) return false;
        if (lm && (!lm->CtC || lm->resetPeriod < L)) return false;      // a split residual reset before the last iteration: the generic kernels' business
        int stripsX = 0, tilesY = 0, G = 0;
        const Variant* V = lm ? select<Op, true>(W, H, cus, stripsX, tilesY, G) : select<Op, false>(W, H, cus, stripsX, tilesY, G);
        if (!V) return false;
        if (!slots) { reserve(W, H, C); if (!slots) return false; HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), ctx.stream)); }
        if (seq > 0xE0000000u || seq + (unsigned)L > 0xE0000000u) {      // tags never repeat: start over on cleared buffers long before the counter wraps
            HIP_CHECK(hipMemsetAsync(slots, 0, slotBytes, ctx.stream)); HIP_CHECK(hipMemsetAsync(box, 0, boxBytes, ctx.stream));
            seq = 2;
        }
        const OcTimeouts tmo = ocTimeouts(timeoutTicks, L, false);
        MoArgs<T> K{W, H, r0, p0, delta, flags, coef, stripsX, tilesY, G, L, seq, slots, box, bad, tmo.later, failAt, tmo.first, lm ? lm->CtC : nullptr, lm ? lm->qTolerance : T(0), (lm || !X) ? hostErr : nullptr, lm ? lm->breakInfo : nullptr};      // (no X: the solver applies the update itself, as for LM)
        {
            ScopedKernel k(ctx, "PCGSolveOnChip");
            Op opc = op;
            void* kargs[] = {(void*)&opc, (void*)&K};
            if (hipLaunchKernel(V->fn, dim3(G), dim3(V->waves * kWave), kargs, 0, ctx.stream) != hipSuccess) { (void)hipGetLastError(); enabled = false; return false; }
        }
        seq += (unsigned)L;
        if (!lm && X) {
            ScopedKernel k(ctx, "PCGLinearUpdate");
            const long n = (long)W * H * C;
            const int grid = (int)std::max<long>(1, std::min<long>((n + kBlock - 1) / kBlock, (long)cus * 8));
            march_applyDelta<T><<<grid, kBlock, 0, ctx.stream>>>(X, delta, n, bad, hostErr);
        }
        launched = true; lastRows = V->rows; lastWaves = V->waves; lastG = G;
        return true;
    }
    bool failedNow() {
        if (!launched) return false;
        launched = false;
        if (__atomic_load_n(hostErr, __ATOMIC_ACQUIRE) == 0) return false;
        failed = true;
        return true;
    }
    bool failedPeek() const { return launched && hostErr && __atomic_load_n(hostErr, __ATOMIC_ACQUIRE) != 0; }      // EnergyOps::onChipFailedPeek
    void rearm(LaunchCtx& ctx) {      // EnergyOps::onChipRearm
        if (!bad) return;
        failed = false; __atomic_store_n(hostErr, 0, __ATOMIC_RELEASE);
        HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), ctx.stream));
    }
    template <class Op> std::string describe(int W, int H, int cus, int L, bool lmv, const char* marchName) const {
        int stripsX = 0, tilesY = 0, G = 0;
        const Variant* V = (enabled && !failed && L > 0) ? (lmv ? select<Op, true>(W, H, cus, stripsX, tilesY, G) : select<Op, false>(W, H, cus, stripsX, tilesY, G)) : nullptr;
        char buf[500];
        if (V) snprintf(buf, sizeof buf, "path=on-chip (march_onchipPcg%s); onchip_rows_per_wave=%d; waves_per_workgroup=%d; wave_tiles=%dx%d of 62 x %d pixels; workgroups=%d of %d CUs; fallback=one launch per PCG iteration (%s)",
                        lmv ? ", LM while lIterations <= residual_reset_period" : "", V->rows, V->waves, stripsX, tilesY, V->rows, G, cus, lmv ? "generic kernels" : marchName);
        else snprintf(buf, sizeof buf, "path=one launch per PCG iteration (%s); why_not_on_chip=%s", lmv ? "generic kernels, LM" : marchName,
                      !enabled ? "switched off" : failed ? "a wait timed out earlier" : "the wave tiles do not fit the CUs");
        return buf;
    }
};

}  // namespace
}  // namespace optamd
