// shape_from_shading: the WHOLE PCG linear solve of one Gauss-Newton / Levenberg-Marquardt step as one persistent launch whose loop state never leaves the chip.
//
// Included by energy_sfs.hip behind the marching kernels (host side: SfsOps::pcgSolveOnChip).  What it replaces: the reference's loop
// `for lIter = 0, lIterations do PCGStep1; PCGStep2; PCGStep3 end` (solverGPUGaussNewton.t:1056-1103) -- three launches and two same-address-atomic sums per
// iteration there, one marching launch per iteration in sfs_pcgMarch -- for images whose loop state fits the register files (the reference's own input is
// 640 x 480, examples/shape_from_shading/src/main.cpp:27-38; BASELINE config 3 is 1024^2).  Same protocol as iw_onchip.h, re-cut for a 5 x 5 coupling:
//   tile    a WAVE holds 64 columns x (R + 4) rows of p and r in registers and owns the 60 x R pixels in the middle: the two-pixel ring around them is held as well
//           and updated by the holder with the same alpha, beta and the same fused operations as by its owner, so the new search direction never travels;
//   march   A p of the owned pixels is one pass over the R + 4 held rows with the expressions of sfs_pcgMarch (row values of the centres of row Y - 1, gather of
//           row Y - 2; DPP shifts for the neighbouring columns, the rows centred on the left / right neighbours summed by those lanes and shifted over as one partial sum per
//           side); what is constant over the solve (dB_I / d{d0, d1, d2}, the flag word, CtC) is re-read per iteration through the caches (read-only, never written
//           while the kernel runs), b sits in LDS;
//   ring    the A p of a tile's two outermost rows / columns goes to a tagged image (one 8-byte {payload, tag} word per float, two per double; relaxed agent-scope
//           stores, no fence), double-buffered by the parity of the iteration; the ring holders pick their pixels' words up INSIDE the wait for the sums;
//   sums    five per iteration (alphaNum, alphaDen, s2, s3 and -- iteration 0 -- sum r^2 / -- later -- Q of the iteration before): every workgroup posts its
//           partial sums as tagged words and adds ALL workgroups' words in workgroup order: the same bits everywhere, so alpha, beta and the q early-out
//           (solver.t:1093-1102) agree on the whole grid without a broadcast.  One grid-wide wait per iteration.
// Levenberg-Marquardt: + CtC p (o.t:2076-2082); Q_k = 1/2 sum delta . (r + b) (:483-485) is formed where iteration k is applied and travels with the sums of
// iteration k + 1 -- exactly the hand-over of the launch-per-iteration loop (solver.hip runSingleKernelLoopLM), so an early-out leaves the reference's delta.
// A split residual reset in the MIDDLE of a linear solve (lIterations > residual_reset_period) is not offered: the host keeps such solves on sfs_pcgMarch.
// Every wait is bounded by the device's wall clock; a time-out raises `bad`, every workgroup leaves the loop at its next sum, nothing is written to delta and the
// host redoes the linear solve with the marching kernels.  The grid must be co-resident (one workgroup per CU): the launcher checks workgroups <= CUs.
#pragma once
#include "onchip_sync.h"

namespace optamd {
namespace {

constexpr int kSoSpan = kWave - 4;
constexpr int kSoMaxG = 256;                  // workgroups (one per CU)
constexpr int kSoNS = 5, kSoNW = 2 * kSoNS;   // sums per iteration; tagged words per workgroup
constexpr int kSoDepth = 2;                   // rows of constants in flight ahead of the row being worked on

template <class T>
struct SfsOcArgs {
    SArgs<T> A;
    const T* r0; const T* p0;           // b = r_0 (solver.t:657)
    const T* CtC;                       // LM
    T* delta;                           // out: sum alpha_k p_k (untouched if a wait timed out)
    int stripsX, tilesY, G, L;
    unsigned tag0;                      // tag of iteration 0 (tags never repeat over the life of the buffers)
    oc_u64* slots;                      // [2][G][10]
    oc_u64* apBox;                      // [2][W * H * sizeof(T) / 4]
    int* bad; long long timeoutTicks; int failAt;
    long long firstTicks;      // bound of the FIRST iteration's wait: the co-residency check (every workgroup has posted its words once it passes), before anything is written
    T qTolerance;
    int* hostErr;                       // LM (the solver applies the update itself): pinned host word a workgroup that gave up raises on its way out; GN: nullptr (sfs_applyDelta tells the host)
    long long* prof;                    // SO_PROFILE builds: [G][8] ticks per phase (wave 0 of every workgroup), else nullptr
    double* lmBreak;                    // pinned {iteration + 1, zeta} of the q early-out (OnChipLm::breakInfo), or nullptr
};

// Development builds (opt_amd/build.py build_variant with SO_PROFILE=1; OPT_AMD_ONCHIP_PROFILE=1): thread 0 of every workgroup accumulates the wall-clock ticks
// (100 MHz) it spends in each phase of an iteration and leaves them in K.prof[workgroup][8].
#ifndef SO_PROFILE
#define SO_PROFILE 0
#endif
#if SO_PROFILE
#define SO_MARK(i) do { if (tid == 0) { const long long t_ = wall_clock64(); soProf[i] += t_ - soPrev; soPrev = t_; } } while (0)
#else
#define SO_MARK(i) do { } while (0)
#endif

template <class T> struct SoRowC { T g0, g1, g2; int fb; };
template <class T> struct SoRow { T v, rk, g0, g1, g2, wr, wc, ws; int ex; };      // a staged row: p, r, dB_I / d{d0, d1, d2}, the three mask multipliers (see the march), `not excluded`

__device__ __forceinline__ float soFma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double soFma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// WAVES: waves per workgroup = per CU (4: one per SIMD, 8: two).  A marching trip is ~350 instructions whatever the row holds, so an iteration costs
// (waves per SIMD) x (R + 4) trips: the launcher picks the (R, WAVES) that minimises it among those whose workgroups fit one per CU.
template <class T, int R, bool LM, int WAVES>
__global__ __launch_bounds__(WAVES * kWave) void sfs_onchipPcg(SfsOcArgs<T> K) {
    constexpr int kSoWaves = WAVES, kSoBlock = WAVES * kWave;
    constexpr int HR = R + 4;                          // held rows: two above and two below the R owned ones
    constexpr int WPS = (int)sizeof(T) / 4;            // tagged words per scalar
    constexpr bool VREG_ARGS = true;
    static_assert(R >= 2, "the ring must come from the adjacent tiles only");
    static_assert(HR % kSoDepth == 0, "the rows requested behind the last trip are the first rows of the next iteration");
    __shared__ double red[kSoNS * kSoWaves];
    __shared__ double TOT[kSoNS + 1];
    __shared__ unsigned W1[kSoMaxG * kSoNW];
    // b = r_0 of the owned pixels (LM: for Q) and -- where the registers are short: double, R >= 8 -- the A p of the owned pixels between the march and the update:
    // [row][thread], conflict-free
    constexpr bool AP_LDS = sizeof(T) * R >= 64;
    __shared__ T bL[(LM ? R : 1) * kSoBlock];
    __shared__ T apL[(AP_LDS ? R : 1) * kSoBlock];
    constexpr bool DL_LDS = sizeof(T) * R >= 80 && WAVES == 8;      // ... and delta itself in the tightest variant (double, R = 10, two waves per SIMD: 1024^2)
    __shared__ T dlL[(DL_LDS ? R : 1) * kSoBlock];
    const SArgs<T>& A = K.A;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = blockIdx.x;      // (wave: uniform, kept in a scalar register)
    const int tile = g * kSoWaves + wave;
    const int sx = tile % K.stripsX, ty = tile / K.stripsX;
    const bool idle = ty >= K.tilesY;                  // (wave-uniform) a wave without a tile: contributes zeros to the sums
    const int x = sx * kSoSpan + lane - 2;
    int yBase = ty * R;                                // first owned row; held row h is image row yBase - 2 + h
    const bool xin = !idle && x >= 0 && x < A.W;
    const bool writer = xin && lane >= 2 && lane < 2 + kSoSpan;
    int xc = min(max(x, 0), A.W - 1);
    const int N = A.W * A.H;
    int* const bad = K.bad;
    const long long to = K.timeoutTicks;
    const T cxc = coefK(A, 0, x, 0);
    const T cxl = dppShift<true>(cxc), cxr = dppShift<false>(cxc);      // (lanes 0 / 63 read 0: they are ring lanes whose row values are never used)

    auto rowIn = [&](int h) { const int y = yBase - 2 + h; return xin && y >= 0 && y < A.H; };
    auto rowIdx = [&](int h) { const int y = yBase - 2 + h; return (y >= 0 && y < A.H && !idle) ? y * A.W + xc : xc; };      // a valid address either way

    // the P(u) coefficient of held row h, wave-uniform: lane h computes it once, a trip reads it from there into scalar registers
    const T cyLane = coefK(A, 1, 0, yBase - 2 + lane);
    auto uniLane = [](T v, int l) -> T {
        if constexpr (sizeof(T) == 8) return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
        else return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    };

    // ---- p_0, r_0 of the held pixels (zeros outside the image); delta = 0 ------------------------------------------------------------------------------------
    T p[HR], r[HR], dl[DL_LDS ? 1 : R], apOwn[AP_LDS ? 1 : R];
#pragma unroll
    for (int h = 0; h < HR; ++h) {
        const int i = rowIdx(h);
        const T pv = K.p0[i], rv = K.r0[i];
        const bool in = rowIn(h);
        p[h] = in ? pv : T(0); r[h] = in ? rv : T(0);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) { if (DL_LDS) dlL[(DL_LDS ? i : 0) * kSoBlock + tid] = 0; else dl[DL_LDS ? 0 : i] = 0; if (!AP_LDS) apOwn[AP_LDS ? 0 : i] = 0; else apL[(AP_LDS ? i : 0) * kSoBlock + tid] = 0; if (LM) bL[(LM ? i : 0) * kSoBlock + tid] = r[i + 2]; }      // b = r_0 (solver.t:657)

    int pixBase = (yBase - 2) * A.W + xc;      // (opaque per iteration below: addresses are recomputed, not kept)
    // The array bases and the three weights are uniform, and the scalar registers are short (lane masks of every predicate live there): kept there, they are spilled to
    // vector lanes and read back -- 32 v_readlane per trip.  As (opaque) vector registers they cost nothing per use.
    const T *g0p = A.g0, *g1p = A.g1, *g2p = A.g2, *ctcp = K.CtC; const uint32_t* flp = A.fl2;
    T wG = A.w_g, wS = A.w_s, wP = A.w_p;
    if (VREG_ARGS) asm volatile("" : "+v"(g0p), "+v"(g1p), "+v"(g2p), "+v"(ctcp), "+v"(flp), "+v"(wG), "+v"(wS), "+v"(wP));
    // the constants of a held row (read-only while the kernel runs: plain cached loads); the first kSoDepth rows of an iteration are requested BEFORE the wait of
    // the iteration before, the others kSoDepth trips ahead of their use
    auto loadRow = [&](int h) {
        SoRowC<T> c;
        const int i = rowIdx(h);
        c.g0 = g0p[i]; c.g1 = g1p[i]; c.g2 = g2p[i]; c.fb = (int)flp[i];
        return c;
    };
    SoRowC<T> cq[kSoDepth];
#pragma unroll
    for (int d = 0; d < kSoDepth; ++d) cq[d] = loadRow(d);
    bool failed = false;
    double accQ = 0;
    T Q0 = 0;                                  // fetchQ before the loop (solver.t:1050): delta = 0, so exactly 0
    const size_t boxStride = (size_t)N * WPS;

#if SO_PROFILE
    long long soProf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, soPrev = wall_clock64();
#endif
    for (int k = 0; k < K.L; ++k) {
        // What is derived from the tile's position (row addresses of six arrays, bounds predicates) is invariant over the solve; hoisted out of this loop it would occupy
        // a hundred registers.  The empty asm makes the sources opaque per iteration, so each use recomputes its two or three instructions.
        asm volatile("" : "+v"(pixBase), "+v"(xc), "+s"(yBase));
        const unsigned tag = K.tag0 + (unsigned)k;
        const int par = (int)(tag & 1u);
        oc_u64* const box = K.apBox + (size_t)par * boxStride;
        oc_u64* const slotPar = K.slots + (size_t)par * K.G * kSoNW;
        if (k == K.failAt && g == 0 && tid == 0) __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool first = k == 0;

        // ---- PCGStep1: A p_k on the owned pixels, with the sums (the expressions of sfs_pcgMarch, in its order) ------------------------------------------------
        double acc = 0, accNum = 0, acc2 = 0, acc3 = 0, accX = 0;
        if (!idle) {
            SoRow<T> R1{}, R2{}, R3{};
            SQ<T> q2{}, q3{};
            T b1 = 0, cy1 = 0, cy2 = 0;
            T ctcQ[2] = {0, 0};      // CtC of the owned rows: requested when the row is staged, used two trips later by its gather
#pragma unroll
            for (int h = 0; h < HR; ++h) {
                asm volatile("" : "+s"(yBase), "+v"(xc));      // (per trip: the row predicates and addresses of all trips are otherwise formed at the top of the iteration and kept -- in scalar registers the kernel does not have)
                const SoRowC<T> c = cq[h % kSoDepth];
                const T ctcNow = ctcQ[h % 2];
                if (LM && h >= 2 && h < R + 2) ctcQ[h % 2] = ctcp[rowIdx(h)];
                cq[h % kSoDepth] = loadRow((h + kSoDepth) % HR);      // (behind the last rows: rows 0 .. kSoDepth - 1 of the next iteration)
                const int Y = yBase - 2 + h;
                // The staged row.  The masks of sfs_pcgMarch -- `interior row centre ? w_g * edge mask : 0`, `regularisation rows on ? w_s : 0` -- are formed ONCE per
                // pixel as multipliers in the solver's precision and travel to the neighbouring columns as such: where the march selects `ok ? m * g : 0` per use, this
                // kernel multiplies by a multiplier that is exactly 0 -- the same products in the same order where the row counts, +-0 where it does not.
                SoRow<T> n;
                {
                    const bool in = rowIn(h);
                    n.v = p[h]; n.rk = r[h];
                    // (the gradient images of a pixel outside the image are whatever the clamped address holds -- finite numbers: they only ever meet a p or a multiplier
                    //  that is exactly 0 there, or end in an output that the `excluded` test below zeroes)
                    n.g0 = c.g0; n.g1 = c.g1; n.g2 = c.g2;
                    const bool ok = in && sfs_interior(A, x, Y);
                    n.wr = ok ? wG * (T)sfsMr(c.fb) : T(0); n.wc = ok ? wG * (T)sfsMc(c.fb) : T(0);
                    n.ws = (ok && (c.fb & kSfsValid)) ? wS : T(0);
                    n.ex = in ? (c.fb & kSfsEx) : 0;
                }
                const T cyN = uniLane(cyLane, h);
                // b(., Y) = g1 v + g0 v(x-1) + g2 v(y-1)                                      (d B_I(c) . v)
                const T vL = dppShift<true>(n.v);
                const T bY = n.g1 * n.v + n.g0 * vL + n.g2 * R1.v;
                // row values at the centres of row Y - 1 (R1); those of the held rows 0 and 1 feed no owned pixel
                SQ<T> qn{};
                if (h >= 2) {
                    const T right = dppShift<false>(b1);
                    qn.gh = R1.wr * (b1 - right);
                    qn.gv = R1.wc * (b1 - bY);
                    const T v1l = dppShift<true>(R1.v), v1r = dppShift<false>(R1.v);
                    T js[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const T c0 = q == 0 ? cxc : q == 1 ? cy1 : T(1), cl = q == 0 ? cxl : q == 1 ? cy1 : T(1), cu = q == 0 ? cxc : q == 1 ? cy2 : T(1),
                                cr = q == 0 ? cxr : q == 1 ? cy1 : T(1), cd = q == 0 ? cxc : q == 1 ? cyN : T(1);
                        T sj = 0;
                        sj += (T(4) * c0) * R1.v; sj += (T(-1) * cl) * v1l; sj += (T(-1) * cu) * R2.v; sj += (T(-1) * cr) * v1r; sj += (T(-1) * cd) * n.v;
                        js[q] = R1.ws * sj;
                    }
                    qn.s0 = js[0]; qn.s1 = js[1]; qn.s2 = js[2];
                }
                // gather of row y = Y - 2 (centre row R2; row values qn at y + 1, q2 at y, q3 at y - 1): held row h - 2, owned row h - 4
                if (h >= 4) {
                    const T ve = R2.v;
                    T s = 0;
                    auto add = [&](T coef, T q) { s += coef * q; };
                    add(wP, wP * ve);      // the fitting row
                    // The rows centred on the pixel itself and on the pixels above / below: as in the march.  The rows centred on the LEFT and RIGHT neighbours are summed by
                    // those lanes -- from their own row values and multipliers, the output pixel's dB_I / P coefficient fetched from it -- and arrive as ONE shifted partial
                    // sum per side: 5 whole-wave shifts instead of 19 (the march shifts every operand).  Same products, another association of the sum.
                    const T g0r = dppShift<false>(R2.g0);
                    add(R2.wr * (R2.g1 - g0r), q2.gh);                     // gh, centre (x, y)
                    add(R1.wr * R1.g2, qn.gh);                             // (x, y+1)
                    add(R2.wc * (R2.g1 - R1.g2), q2.gv);                   // gv, centre (x, y)
                    add(R1.wc * R1.g2, qn.gv);                             // (x, y+1)
                    add(-(R3.wc * R2.g1), q3.gv);                          // (x, y-1)
                    T sReg = 0;      // (the regularisation rows of the column in a chain of their own: four independent chains per pixel instead of one of 26 dependent operations)
                    auto reg = [&](T ws, T w4, T a0, T a1, T a2) {
                        const T wgt = ws * w4;
                        sReg += (wgt * cxc) * a0; sReg += (wgt * cy2) * a1; sReg += (wgt * T(1)) * a2;
                    };
                    reg(R2.ws, T(4), q2.s0, q2.s1, q2.s2);
                    reg(R1.ws, T(-1), qn.s0, qn.s1, qn.s2);
                    reg(R3.ws, T(-1), q3.s0, q3.s1, q3.s2);
                    s += sReg;
                    {
                        const T wgt = R2.ws * T(-1), wy = wgt * cy2, w1 = wgt * T(1);
                        // for the pixel on the LEFT, whose right-hand neighbour this lane is: rows (x+1, y), (x+1, y-1) of its gather
                        T tR = 0;
                        tR += (R2.wr * R2.g0) * q2.gh; tR += (R2.wc * R2.g0) * q2.gv; tR += -(R3.wc * R2.g0) * q3.gv;
                        tR += (wgt * cxl) * q2.s0; tR += wy * q2.s1; tR += w1 * q2.s2;
                        // for the pixel on the RIGHT: rows (x-1, y), (x-1, y+1) of its gather (its own dB_I / d d1 at rows y, d d2 at row y + 1 multiply them)
                        const T g1R = dppShift<false>(R2.g1), g2R1 = dppShift<false>(R1.g2);
                        T tL = 0;
                        tL += -(R2.wr * g1R) * q2.gh; tL += -(R1.wr * g2R1) * qn.gh;
                        tL += (wgt * cxr) * q2.s0; tL += wy * q2.s1; tL += w1 * q2.s2;
                        s += dppShift<false>(tR);
                        s += dppShift<true>(tL);
                    }
                    if (LM) s += ctcNow * ve;
                    if (!R2.ex) s = 0;
                    if (AP_LDS) apL[(AP_LDS && h >= 4 ? h - 4 : 0) * kSoBlock + tid] = s; else apOwn[!AP_LDS && h >= 4 ? h - 4 : 0] = s;
                    if (writer && Y - 2 < A.H) {
                        acc += (double)(ve * s);
                        const T rk = R2.rk;
                        const T zk = first ? ve : rk;                                          // iteration 0: alphaNumerator_0 = r_0 . p_0 (the reference's start)
                        accNum += (double)(zk * rk); acc2 += (double)(rk * s); acc3 += (double)(s * s);
                        if (first) accX += (double)(rk * rk);
                        // the tile's two outermost rows / columns: to the tagged image, for whoever holds them as ring
                        if (h - 4 < 2 || h - 4 >= R - 2 || lane < 4 || lane >= kWave - 4) {
                            const int i = pixBase + (h - 2) * A.W;
                            if constexpr (WPS == 1) ocStore(box + i, tag, __float_as_uint((float)s));
                            else { const oc_u64 b = (oc_u64)__double_as_longlong((double)s); ocStore(box + 2 * (size_t)i, tag, (unsigned)b); ocStore(box + 2 * (size_t)i + 1, tag, (unsigned)(b >> 32)); }
                        }
                    }
                }
                R3 = R2; R2 = R1; R1 = n; q3 = q2; q2 = qn; b1 = bY; cy2 = cy1; cy1 = cyN;
                __builtin_amdgcn_sched_barrier(0);      // one row per scheduling region: left to itself the scheduler interleaves the unrolled rows until their temporaries fill the register budget
            }
        }
        SO_MARK(0);      // march
        if (!first) accX = accQ;      // Q of the iteration before (LM; 0 otherwise)
        asm volatile("" : "+v"(pixBase), "+v"(xc), "+s"(yBase));      // (the row predicates of the march are not kept for the wait: recomputed there)

        // ---- the grid-wide sums; the ring's A p is collected inside the wait ------------------------------------------------------------------------------------
        {
            double v5[kSoNS] = {accNum, acc, acc2, acc3, accX};
#pragma unroll
            for (int q = 0; q < kSoNS; ++q) { v5[q] = ocWaveSum63(v5[q]); if (lane == kWave - 1) red[q * kSoWaves + wave] = v5[q]; }
        }
        SO_MARK(1);      // wave sums
        __syncthreads();
        SO_MARK(2);      // barrier: the slowest wave's march
        if (tid < kSoNW) {
            double s = 0;
            for (int w = 0; w < kSoWaves; ++w) s += red[(tid >> 1) * kSoWaves + w];
            const oc_u64 b = (oc_u64)__double_as_longlong(s);
            ocStore(slotPar + (size_t)g * kSoNW + tid, tag, (tid & 1) ? (unsigned)(b >> 32) : (unsigned)b);
        }
        // ---- ONE wait: the ring's A p (held pixels inside the image that this wave does not own; posted before their owners' sums) and every workgroup's sums are
        // requested together, re-requested until all carry this iteration's tag
        T ring[HR];
        {
            // (measured: a word-major layout that lets every wave request "its" sum directly -- no staging -- makes eight workgroups post into one 64-byte line: the wait
            //  grows from 3.5 to 5.6 us.  Workgroup-major words, thread i requests word i, the words are regrouped through LDS.)
            constexpr int kPer = (kSoMaxG * kSoNW + kSoBlock - 1) / kSoBlock;
            oc_u64 w[kPer];
            const int nW = K.G * kSoNW;
            const bool lastIt = k + 1 == K.L;      // (after the last iteration only delta survives: nobody needs the ring)
            auto need = [&](int h) { return !lastIt && rowIn(h) && !(writer && h >= 2 && h < R + 2); };
            oc_u64 rw[HR][WPS];
            bool sumsOk = false, ringOk = false;
            // requests and checks apart: the first round asks for everything at once; a later round asks again only for what has not arrived (the ring words are
            // posted before their owners' sums and are normally there by then: the re-requests are the ten words of the sums)
            auto askSums = [&]() {
#pragma unroll
                for (int u = 0; u < kPer; ++u) { const int i = tid + u * kSoBlock; w[u] = ocLoad(slotPar + (i < nW ? i : tid % nW)); }
            };
            auto askRing = [&]() {
#pragma unroll
                for (int h = 0; h < HR; ++h) {
#pragma unroll
                    for (int q = 0; q < WPS; ++q) rw[h][q] = (oc_u64)tag << 32;
                    if (need(h)) {
                        const size_t i = (size_t)(pixBase + h * A.W) * WPS;
#pragma unroll
                        for (int q = 0; q < WPS; ++q) rw[h][q] = ocLoad(box + i + q);
                    }
                }
            };
            auto check = [&]() {
                if (!sumsOk) {
                    bool ok = true;
#pragma unroll
                    for (int u = 0; u < kPer; ++u) { const int i = tid + u * kSoBlock; ok = ok && (i >= nW || (unsigned)(w[u] >> 32) == tag); }
                    sumsOk = ok;
                }
                if (!ringOk) {
                    bool ok = true;
#pragma unroll
                    for (int h = 0; h < HR; ++h) {
#pragma unroll
                        for (int q = 0; q < WPS; ++q) ok = ok && (unsigned)(rw[h][q] >> 32) == tag;
                    }
                    ringOk = ok;
                }
                return sumsOk && ringOk;
            };
            askSums(); askRing();
#if SO_PROFILE
            const bool firstOk = check();
            if (tid == 0) { const long long t_ = wall_clock64(); soProf[7] += t_ - soPrev; }      // (slot 7: the first round alone; slot 6: rounds after it, counted by thread 0)
            if (!firstOk) {
#else
            if (!check()) {
#endif
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                for (;;) {
                    __builtin_amdgcn_s_sleep(1);
                    if (!sumsOk) askSums();
                    if (!ringOk) askRing();
#if SO_PROFILE
                    if (tid == 0) soProf[6] += 1;
#endif
                    if (check()) break;
                    if ((++spins & 31u) == 0) {
                        if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        if (wall_clock64() - t0 > (k == 0 ? K.firstTicks : to)) { __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < HR; ++h) {
                if constexpr (WPS == 1) ring[h] = __uint_as_float((unsigned)rw[h][0]);
                else ring[h] = __longlong_as_double((long long)((rw[h][WPS - 1] << 32) | (rw[h][0] & 0xffffffffull)));
            }
            SO_MARK(3);      // the wait: ring + sums words
#pragma unroll
            for (int u = 0; u < kPer; ++u) { const int i = tid + u * kSoBlock; if (i < nW) W1[i] = (unsigned)w[u]; }
            __syncthreads();
            // Every workgroup adds all workgroups' words in the same order: wave q (the fifth sum: wave 0 again) takes sum q, a lane the workgroups lane, lane + 64,
            // lane + 128, lane + 192 in that order, then the wave's DPP tree -- the same association everywhere, so the same bits.
#pragma unroll
            for (int pass = 0; pass < (kSoNS + kSoWaves - 1) / kSoWaves; ++pass) {
                const int q = wave + pass * kSoWaves;
                if (q < kSoNS) {
                    double sacc = 0;
#pragma unroll
                    for (int c = 0; c < kSoMaxG / kWave; ++c) {
                        const int m = lane + c * kWave;
                        const double v = m < K.G ? ocJoin(W1[m * kSoNW + 2 * q], W1[m * kSoNW + 2 * q + 1]) : 0.0;
                        sacc += v;
                    }
                    sacc = ocWaveSum63(sacc);
                    if (lane == kWave - 1) TOT[q] = sacc;
                }
            }
            if (tid == 0) reinterpret_cast<int*>(TOT + kSoNS)[0] = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
        SO_MARK(4);      // sum over workgroups
        const double aNumD = TOT[0], aDenD = TOT[1], s2 = TOT[2], s3 = TOT[3], xD = TOT[4];
        if (reinterpret_cast<const int*>(TOT + kSoNS)[0]) { failed = true; break; }      // uniform over the workgroup: a wait timed out somewhere
        if (LM && !first) {      // the q early-out of iteration k - 1 (solver.t:1093-1102): nothing of iteration k has been applied yet
            const T Q1 = (T)xD;
            const T zeta = T(k) * (Q1 - Q0) / Q1;
            if (zeta < K.qTolerance) { if (K.lmBreak && blockIdx.x == 0 && tid == 0) { K.lmBreak[1] = (double)zeta; K.lmBreak[0] = (double)(k + 1); } break; }
            Q0 = Q1;
        }
        // the scalars of sfs_pcgMarch's prologue (solver.t:456-459, 544-547 guards; beta numerator by expansion, clamped like the direct sum it replaces)
        const T aNum = (T)aNumD, aDen = (T)aDenD;
        const T alpha = (aDen > T(0)) ? aNum / aDen : T(0);
        const double rr = first ? xD : aNumD;
        const double bNumD = fmax(rr - 2.0 * (double)alpha * s2 + (double)alpha * (double)alpha * s3, 0.0);
        const T beta = (aNum > T(0)) ? (T)bNumD / aNum : T(0);
        const bool last = k + 1 == K.L;

        // ---- PCGStep2 + PCGStep3 (z = r: this energy does not precondition after the start): delta += alpha p;  r -= alpha A p;  p = r + beta p -- on the owned
        // pixels and, with the same fused operations, on the ring (after the last iteration only delta survives)
        accQ = 0;
#pragma unroll
        for (int h = 0; h < HR; ++h) {
            const bool ownRow = h >= 2 && h < R + 2;
            const T apv = ownRow ? (writer ? (AP_LDS ? apL[(AP_LDS && ownRow ? h - 2 : 0) * kSoBlock + tid] : apOwn[!AP_LDS && ownRow ? h - 2 : 0]) : ring[h]) : ring[h];
            T dNew = 0;
            if (ownRow) {
                dNew = soFma(alpha, p[h], DL_LDS ? dlL[(DL_LDS && ownRow ? h - 2 : 0) * kSoBlock + tid] : dl[!DL_LDS && ownRow ? h - 2 : 0]);
                if (DL_LDS) dlL[(DL_LDS && ownRow ? h - 2 : 0) * kSoBlock + tid] = dNew; else dl[!DL_LDS && ownRow ? h - 2 : 0] = dNew;
            }
            if (!last) {
                r[h] = soFma(-alpha, apv, r[h]);
                if (LM && ownRow && writer && yBase + (h - 2) < A.H) accQ += (double)(T(0.5) * (dNew * (r[h] + bL[(LM && ownRow ? h - 2 : 0) * kSoBlock + tid])));      // solver.t:483-485
                p[h] = soFma(beta, p[h], r[h]);
            }
        }
        SO_MARK(5);      // update
    }
#if SO_PROFILE
    if (tid == 0 && K.prof) { for (int i = 0; i < 8; ++i) K.prof[(long)g * 8 + i] = soProf[i]; }
#endif
    if (failed && tid == 0 && K.hostErr) __hip_atomic_store(K.hostErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!failed && writer) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int y = yBase + i;
            if (y < A.H) K.delta[y * A.W + x] = DL_LDS ? dlL[(DL_LDS ? i : 0) * kSoBlock + tid] : dl[DL_LDS ? 0 : i];
        }
    }
}

// PCGLinearUpdate X += delta (solver.t:552-557) behind the on-chip Gauss-Newton solve -- unless a wait timed out: then the unknowns stay untouched and the host is told
template <class T>
__global__ __launch_bounds__(kBlock) void sfs_applyDelta(T* __restrict__ X, const T* __restrict__ delta, long N, const int* __restrict__ bad, int* hostErr) {
    if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(hostErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) X[i] = X[i] + delta[i];
}

}  // namespace
}  // namespace optamd
