// image_warping, unit lattice, Gauss-Newton and Levenberg-Marquardt: the WHOLE PCG linear solve as one persistent launch whose loop state never leaves the chip.
//
// Included by energy_image_warping.hip (host side: ImageWarpingOps::pcgSolveOnChip).  What it replaces: the reference's loop
// `for lIter = 0, lIterations do PCGStep1; PCGStep2; PCGStep3 end` (solverGPUGaussNewton.t:1056-1092) -- three launches and two same-address-atomic
// sums per iteration there, one streaming launch per iteration in iw_pcgIter2 -- for problems whose state fits the register files and LDS of the
// chip (<= 8192 pixels per CU: 2 M pixels on 256 CUs; the reference's own inputs are 512^2 and 640x480, and 1/8 of the metric's 4096^2 is 4096x512).
// The reference's precedent for an on-chip solve is its block-local comparator (examples/poisson_image_editing/src/PatchSolverWarping.cu:67-196:
// one patch per block, state in shared memory); this kernel is a GLOBAL solve -- the same iterates as the streaming loop -- that keeps
//   p, r            in registers (a lane owns ROWS consecutive rows of one image column),
//   A p             in registers (ROWS <= 8) or LDS (ROWS = 16, 96 KB),
//   delta           in registers, or -- ROWS = 16 -- in the solver's delta vector, read-modify-written once per iteration (the 3 MB per XCD stay in its L2),
//   cos / sin, flag byte of every pixel in registers,
// and synchronises the grid ONCE per iteration with 8-byte {payload, tag} words (one relaxed agent-scope store each, no fences, no cache write-backs;
// MI355X_MICROARCH.md "handoff-1to1" / "allgather"; measured in tools/microbench_gridsync.hip):
//   halo   a workgroup's tile is 256 x 2 ROWS pixels (8 waves: 4 across, 2 down).  Every wave keeps p and r of the one-pixel ring around its pixels itself
//          and receives only the A p of those pixels -- through LDS inside a workgroup, through per-tile inboxes in global memory between workgroups --
//          posted TOGETHER with the partial sums, so the hand-over rides on the wait for the sums (see the kernel's header);
//   sum    the four sums of the iteration (alphaDen = p.Ap, alphaNum = sum M r^2, s2 = sum M r.Ap, s3 = sum M Ap^2; beta by expansion as in iw_pcgIter2,
//          energy.h PcgIterArgs) as a two-level tree: 16 workgroups per group, group totals posted by the group's first workgroup, every workgroup adds the
//          group totals in group order -- the same bits everywhere, so alpha and beta agree on the whole grid without a broadcast.
// Every wait is bounded by the device's wall clock; a time-out raises K.S.bad, every workgroup leaves the loop at its next sum, nothing is applied to the
// unknowns (iw_applyDelta checks the flag) and the host falls back to the streaming loop.  The grid must be co-resident (one workgroup per CU): the launcher
// checks tiles <= CUs x occupancy.
#pragma once
#include "iw_device.h"
#include "onchip_sync.h"

namespace optamd {
namespace {

constexpr int kOcBlock = 512, kOcWavesX = 4, kOcWavesY = 2, kOcWaves = kOcBlock / kWave, kOcTileW = kOcWavesX * kWave;
constexpr int kOcGroup = 16;                  // workgroups per first-level group of the grid-wide sum
constexpr int kOcMaxTiles = 256;              // 16 groups of 16

struct OnchipSync {
    oc_u64* slots;          // [2][G][8]: a workgroup's four double sums as 8 tagged halves
    oc_u64* groupSlots;     // [2][ceil(G / 16)][8]
    oc_u64* inbox;          // [2][G][4 sides][stride]: edge rows / columns of p from the four neighbouring tiles
    int* bad;               // device word: some wait timed out
    int* hostErr;           // pinned host word, set by iw_applyDelta when `bad` is
    long stride;            // words per (tile, side): 3 * kOcTileW scalars
};
// Row slabs (one rank per GPU): the links of this rank's kernel to the other ranks' (include/OptAmd.h OptAmd_OnChipLinks; world <= 1: none).
struct OcLinks {
    oc_u64* mailDst[16]; const oc_u64* mailMine;
    int world, rank, slots, slotStride, rankStride; unsigned seq0;
    oc_u64 *edgeSendUp, *edgeSendDown; const oc_u64 *edgeRecvUp, *edgeRecvDown; long edgeParityStride;
};
template <class T>
struct OnchipArgs {
    int W, H, tilesX, tilesY, G;        // H: rows of the arrays (a slab's include its ghost rows)
    int yBegin, yEnd;                   // the rows the tiles cover: [0, H) on one GPU, the slab's owned rows (a multiple of the tile height) otherwise
    OcLinks links;
    const T* r0; const T* p0;           // solver layout: [O.x O.y] x N, then [a] x N
    const T* Angle; const uint8_t* flags;
    T* delta;                           // out: sum alpha_k p_k
    T w_fit, w_reg;
    int L; unsigned tag0;               // iterations; tag of iteration 0 (tags never repeat over the life of the buffers)
    int flat;                           // 1: every workgroup reads every workgroup's slot (small grids); 0: two-level tree
    OnchipSync S;
    double* trace;                      // [L][4] = alphaNum, alphaDen, s2, s3 of every iteration (written by workgroup 0), or nullptr
    long long timeoutTicks;
    long long firstTicks;               // bound of the waits of the FIRST phase: passing it proves that every workgroup of the grid is resident (each has posted its words), so it is the
                                        // co-residency check -- short (10 ms), before anything has been written; row slabs: = timeoutTicks (the first sum waits for the other ranks' launches)
    long long* prof;                    // OC_PROFILE builds: [G][8] ticks per phase, else nullptr
    int failAt;                         // test hook (OPT_AMD_ONCHIP_FAIL_AT): workgroup 0 raises `bad` in this iteration as a timed-out wait would; -1: never
    // Levenberg-Marquardt variants (LMV): the scalars of PCGFinalizeDiagonal (solver.t:631-664), the q early-out and the residual reset period (:1077-1102)
    T lmRadius, lmMin, lmMax, qTolerance; int resetPeriod;      // (LMV: `trace` is the pinned {iteration + 1, zeta} word of the q early-out instead, OnChipLm::breakInfo -- the LM variants are never traced)
};

// one scalar of the halo as tagged words: a float is one word, a double two
template <bool SYS = false> __device__ __forceinline__ void ocSend(oc_u64* box, int idx, float v, unsigned tag) { ocStore<SYS>(box + idx, tag, __float_as_uint(v)); }
template <bool SYS = false> __device__ __forceinline__ void ocSend(oc_u64* box, int idx, double v, unsigned tag) {
    const oc_u64 b = (oc_u64)__double_as_longlong(v);
    ocStore<SYS>(box + 2 * idx, tag, (unsigned)b); ocStore<SYS>(box + 2 * idx + 1, tag, (unsigned)(b >> 32));
}
__device__ __forceinline__ void ocRecv(const oc_u64* box, int idx, unsigned tag, int* bad, long long to, float& v) { v = __uint_as_float(ocAwait(box + idx, tag, bad, to)); }
__device__ __forceinline__ void ocRecv(const oc_u64* box, int idx, unsigned tag, int* bad, long long to, double& v) {
    const unsigned lo = ocAwait(box + 2 * idx, tag, bad, to), hi = ocAwait(box + 2 * idx + 1, tag, bad, to);
    v = __longlong_as_double((long long)(((oc_u64)hi << 32) | lo));
}
// Three scalars of one halo pixel at once: all requests are in flight together (one fabric round trip when the words are already there, not three).
template <bool SYS = false> __device__ __forceinline__ void ocRecv3(const oc_u64* box, int i0, int i1, int i2, unsigned tag, int* bad, long long to, float (&v)[3]) {
    const oc_u64 *p0 = box + i0, *p1 = box + i1, *p2 = box + i2;
    oc_u64 a = ocLoad<SYS>(p0), b = ocLoad<SYS>(p1), c = ocLoad<SYS>(p2);
    if ((unsigned)(a >> 32) != tag || (unsigned)(b >> 32) != tag || (unsigned)(c >> 32) != tag) {
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
            a = ocLoad<SYS>(p0); b = ocLoad<SYS>(p1); c = ocLoad<SYS>(p2);
            if ((unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag && (unsigned)(c >> 32) == tag) break;
            if ((++spins & 31u) == 0) {
                if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (wall_clock64() - t0 > to) { __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
    }
    v[0] = __uint_as_float((unsigned)a); v[1] = __uint_as_float((unsigned)b); v[2] = __uint_as_float((unsigned)c);
}
template <bool SYS = false> __device__ __forceinline__ void ocRecv3(const oc_u64* box, int i0, int i1, int i2, unsigned tag, int* bad, long long to, double (&v)[3]) {
    const oc_u64* q[6] = {box + 2 * i0, box + 2 * i0 + 1, box + 2 * i1, box + 2 * i1 + 1, box + 2 * i2, box + 2 * i2 + 1};
    oc_u64 w[6];
    auto fetch = [&]() { bool ok = true; for (int i = 0; i < 6; ++i) w[i] = ocLoad<SYS>(q[i]); for (int i = 0; i < 6; ++i) ok = ok && (unsigned)(w[i] >> 32) == tag; return ok; };
    if (!fetch()) {
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
            if (fetch()) break;
            if ((++spins & 31u) == 0) {
                if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (wall_clock64() - t0 > to) { __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
    }
    for (int i = 0; i < 3; ++i) v[i] = __longlong_as_double((long long)((w[2 * i + 1] << 32) | (w[2 * i] & 0xffffffffull)));
}

// Whole-wave shifts that KEEP `old` in the lane whose source lies outside the wave (bound_ctrl off): lane 0 of fromLeft / lane 63 of fromRight receive the
// halo value the caller put there, every other lane its neighbour's register -- the wave-edge column costs no extra instruction.
__device__ __forceinline__ int ocFromLeft(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ int ocFromRight(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ float ocFromLeft(float old, float v) { return __int_as_float(ocFromLeft(__float_as_int(old), __float_as_int(v))); }
__device__ __forceinline__ float ocFromRight(float old, float v) { return __int_as_float(ocFromRight(__float_as_int(old), __float_as_int(v))); }
__device__ __forceinline__ double ocFromLeft(double old, double v) {
    return __hiloint2double(ocFromLeft(__double2hiint(old), __double2hiint(v)), ocFromLeft(__double2loint(old), __double2loint(v)));
}
__device__ __forceinline__ double ocFromRight(double old, double v) {
    return __hiloint2double(ocFromRight(__double2hiint(old), __double2hiint(v)), ocFromRight(__double2loint(old), __double2loint(v)));
}

template <class T> struct __attribute__((aligned(16))) OcH4 { T v[4]; };      // {ox, oy, a, -} of p, r or A p, or {cos, sin, on, flag byte} of one halo pixel

__device__ __forceinline__ float ocFma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double ocFma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// LDS carve-up (bytes), shared by the kernel and the launcher
constexpr int kOcSumsMax = 6;                 // capacity of the per-phase sums (Gauss-Newton 4, Levenberg-Marquardt 5)
template <class T> struct OcLds {
    static constexpr size_t ap(int rows, bool apLds) { return apLds ? (size_t)rows * 3 * kOcBlock * sizeof(T) : 0; }
    static constexpr size_t row() { return (size_t)kOcWaves * 2 * 3 * kWave * sizeof(T); }
    static constexpr size_t rows3(bool apLds) { return (apLds ? 2 : 3) * row(); }      // p and r of the halo rows; their A p only when it cannot be read from apL
    static constexpr size_t side(int rows) { return (size_t)kOcWaves * rows * 2 * sizeof(OcH4<T>); }
    static constexpr size_t tail() { return (kOcSumsMax * kOcWaves + kOcGroup * kOcSumsMax + 8) * sizeof(double) + (kOcMaxTiles * 2 * kOcSumsMax + kOcGroup * 8) * sizeof(unsigned) + 32 * sizeof(T) + 16; }
    // Levenberg-Marquardt: b = r_0 of the lane's pixels ([row][component][thread], like apL), delta of the halo rows and of the halo columns
    static constexpr size_t lm(int rows) { return ap(rows, true) + row() + side(rows); }
    // cos / sin of the lane's LAST csRows rows live in LDS instead of registers where that relieves a variant that spills and the LDS has the room: the LM ROWS = 8 variant
    // (round 6: 124 -> 68-76 B of scratch per lane; a row's pair is read back once per stencil pass): [row][cos, sin][thread], behind everything else.  (ROWS = 16 has no room: its
    // A p fills 96 of the CU's 160 KB and the rest is taken to within 6 KB; its 40 B of scratch stay -- ~13 scratch operations per iteration of ~2000 VALU instructions.)
    static constexpr int csRows(int rows, bool apLds, bool lmv) { return (sizeof(T) == 4 && rows == 8 && lmv && !apLds) ? 8 : 0; }
    static constexpr size_t base(int rows, bool apLds, bool lmv) { return ap(rows, apLds) + rows3(apLds) + 5 * side(rows) + tail() + (lmv ? lm(rows) : 0); }
    static constexpr size_t total(int rows, bool apLds, bool lmv = false) { return base(rows, apLds, lmv) + (size_t)csRows(rows, apLds, lmv) * 2 * kOcBlock * sizeof(T); }
};

// Development builds (opt_amd/build.py build_variant with OC_PROFILE=1; tools/onchip_bench.py under OPT_AMD_ONCHIP_PROFILE=1): thread 0 of every workgroup
// accumulates the wall-clock ticks (100 MHz) it spends in each phase of an iteration and leaves them in K.prof[workgroup][8].
#ifndef OC_PROFILE
#define OC_PROFILE 0
#endif
#if OC_PROFILE
#define OC_MARK(i) do { if (lane == 0) { const long long t_ = wall_clock64(); ocProf[wave * 16 + (i)] += t_ - ocPrev; ocPrev = t_; } } while (0)
#else
#define OC_MARK(i) do { } while (0)
#endif

// ONE grid-wide wait per iteration.  A tile keeps p and r of its own pixels AND of the one-pixel ring around every wave (rows above / below: registers, lane-aligned;
// columns left / right: LDS, one lane per halo pixel).  After the stencil a wave hands the A p of its edge pixels to whoever holds them as halo -- the neighbouring
// wave through LDS, the neighbouring tile through its inbox -- together with the workgroup's partial sums; after the one wait everybody applies PCGStep2 / PCGStep3
// to its own pixels and to its halo copies with the same alpha, beta and the same explicitly fused operations: owner and halo holder get the same bits, and the
// new search direction never has to travel.
//
// LMV: the Levenberg-Marquardt loop (solverGPUGaussNewton.t:1056-1103 with the LM branches) in the same protocol:
//   * A = J^T J + diag(CtC) (o.t:2076-2082); CtC and the LM preconditioner of PCGFinalizeDiagonal (:631-664) are 15-entry tables indexed by the flag byte, as in
//     iw_pcgIter2<.., LM> (on a unit lattice diag(J^T J) is a function of the flag byte, and SSq = guardedInvert(diag) never changes);
//   * Q_k = 1/2 sum delta . (r + b) (:483-485) is formed where PCGStep2 updates delta and r -- behind the wait of iteration k -- and travels with the sums of
//     iteration k + 1: the zeta test of iteration k (:1093-1102) is decided by EVERY workgroup from the same five totals at the wait of iteration k + 1, before
//     anything of iteration k + 1 has been applied, so an early-out leaves exactly the reference's delta (its last PCGStep3 is dead);
//   * every residual_reset_period-th iteration ends with the split PCGStep2 (:1077-1083, 491-534): delta += alpha p, then a SECOND stencil pass A delta with its own
//     hand-over and grid-wide wait (phase B), r = b - A delta, z = M r, beta = sum z.r / alphaNumerator, Q directly.  Halo holders keep delta of their ring
//     pixels as well, and the second pass hands them the new r of the edge pixels in the words the first pass uses for A p.
// Phases (not iterations) number the tags and select the parity of the double-buffered boxes: a workgroup can pass the wait of phase n + 1 only after every
// workgroup has read its phase-n words.
template <class T, int ROWS, bool AP_LDS, bool DELTA_GLB, bool LMV = false>
__global__ __launch_bounds__(kOcBlock, 2) void iw_onchipPcg(OnchipArgs<T> K) {
    static_assert(2 * ROWS <= kWave, "one lane per halo pixel of the two side columns");
    static_assert(!LMV || (!AP_LDS && !DELTA_GLB), "the LM variants keep A p and delta in registers");
    constexpr int NS = LMV ? 5 : 4, NW = 2 * NS;      // sums per phase; tagged words per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char ocLds[];
    T* apL = reinterpret_cast<T*>(ocLds);                                                       // [ROWS * 3][512]: conflict-free [row][component][thread]
    T* rowP = reinterpret_cast<T*>(ocLds + OcLds<T>::ap(ROWS, AP_LDS));                         // [wave][0 = above, 1 = below][3][64]: p of the halo rows (lane-aligned)
    T* rowR = rowP + kOcWaves * 2 * 3 * kWave;                                                  // their r
    T* rowA = rowR + kOcWaves * 2 * 3 * kWave;                                                  // their A p, written by the neighbouring wave (only without apL: with it the holder reads the owner's A p there)
    OcH4<T>* sideP = reinterpret_cast<OcH4<T>*>(reinterpret_cast<unsigned char*>(rowP) + OcLds<T>::rows3(AP_LDS));      // [wave][row][0 = left, 1 = right]: p of the halo columns
    OcH4<T>* sideR = sideP + kOcWaves * ROWS * 2;                                               // r of the same pixels
    OcH4<T>* sideC = sideR + kOcWaves * ROWS * 2;                                               // their cos, sin, on, flag byte (constant over the solve)
    OcH4<T>* sideA = sideC + kOcWaves * ROWS * 2;                                               // their A p, written by the neighbouring wave's edge lane
    OcH4<T>* stageA = sideA + kOcWaves * ROWS * 2;                                              // [wave][row][0 = lane 0's, 1 = lane 63's]: edge A p on its way to another tile
    double* red = reinterpret_cast<double*>(stageA + kOcWaves * ROWS * 2);                      // [sums][waves]
    double* GS = red + kOcSumsMax * kOcWaves;                                                   // [groups][sums]
    double* TOT = GS + kOcGroup * kOcSumsMax;                                                   // [sums] + the bad flag (at TOT[6])
    unsigned* W1 = reinterpret_cast<unsigned*>(TOT + 8);                                        // [<= 256 workgroups][NW]
    unsigned* W2 = W1 + kOcMaxTiles * 2 * kOcSumsMax;                                           // [<= 16 groups][8]
    T* mTab = reinterpret_cast<T*>(W2 + kOcGroup * 8);                                          // guardedInvert(diag J^T J) by flag byte, as in iw_pcgIter2 (PRE == 3); LM: the LM preconditioner
    T* cTab = mTab + 16;                                                                        // LM: CtC by flag byte
    // Levenberg-Marquardt only (behind everything else, so the Gauss-Newton layout does not move)
    T* bL = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(mTab + 32) + 16);             // [ROWS * 3][512]: b = r_0 of the lane's pixels
    T* rowD = bL + (LMV ? ROWS * 3 * kOcBlock : 0);                                             // delta of the halo rows
    OcH4<T>* sideD = reinterpret_cast<OcH4<T>*>(rowD + kOcWaves * 2 * 3 * kWave);               // delta of the halo columns

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6, wx = wave & (kOcWavesX - 1), wy = wave / kOcWavesX;
    const int g = blockIdx.x, tx = g % K.tilesX, ty = g / K.tilesX;
    const int x0 = tx * kOcTileW + wx * kWave, x = x0 + lane, yBase = K.yBegin + (ty * kOcWavesY + wy) * ROWS;
    const long N = (long)K.W * K.H;
    const bool xin = x < K.W;
    const T w2 = K.w_reg * K.w_reg, wf2 = K.w_fit * K.w_fit;
    int* const bad = K.S.bad;
    // A launch enqueued behind one whose wait timed out (Opt_ProblemSolve enqueues several Gauss-Newton steps before it reads anything back): nothing to do -- the flag is
    // sticky until the host re-arms the path, PCGLinearUpdate checks it too, and waiting for peers again would cost the first-phase bound per launch.
    if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;

    if (tid < 15) {      // the table of iw_pcgIter2: same accumulation order as iw_evalJTF, so the entries are the solver's preconditioner values bit for bit
        const int t = tid, cnt = t < 10 ? t % 5 : t - 10;
        const T w = K.w_reg;
        T d = 0;
        if (t < 10) { for (int n = 0; n < cnt; ++n) d += w * w + w * w; if (t >= 5) d += K.w_fit * K.w_fit; }
        else for (int n = 0; n < cnt; ++n) d += (w * T(1)) * (w * T(1));
        const T sq = T(1) + sqrt(d);
        const T gi = T(1) / (sq * sq);
        if (LMV) {      // k_finalizeDiagonal (solver.t:631-664) on the table, exactly as iw_pcgIter2<.., LM = true> forms it
            const T radius = K.lmRadius, unclamped = d * (T(1) / radius), clampMul = (T(1) / gi) / radius;
            const T c = fmin(fmax(unclamped, K.lmMin * clampMul), K.lmMax * clampMul);
            cTab[t] = c; mTab[t] = T(1) / (c + radius * unclamped);
        } else mTab[t] = gi;
    }
    // p_0, r_0, cos, sin, activity and flag byte of a pixel that may lie outside the image (then: zeros, inactive)
    // (own pixels: the rows the tiles cover, [yBegin, yEnd); halo pixels: any row of the arrays -- a slab's ghost rows hold the neighbouring rank's pixels, and the
    // flag byte of a ghost row beyond the global image says "does not exist")
    auto loadPixel = [&](int xx, int yy, T (&pp)[3], T (&rr)[3], T& c, T& s, T& on, unsigned& f, bool own = false) {
        const bool ok = xx >= 0 && xx < K.W && yy >= 0 && yy < (own ? K.yEnd : K.H);
        const long i = ok ? (long)yy * K.W + xx : 0;
        f = ok ? (unsigned)K.flags[i] : 0u;
        const V2<T> po = ((const V2<T>*)K.p0)[i], ro = ((const V2<T>*)K.r0)[i];
        const T pa = K.p0[2 * N + i], ra = K.r0[2 * N + i];
        pp[0] = ok ? po.x : T(0); pp[1] = ok ? po.y : T(0); pp[2] = ok ? pa : T(0);
        rr[0] = ok ? ro.x : T(0); rr[1] = ok ? ro.y : T(0); rr[2] = ok ? ra : T(0);
        sincosT(K.Angle[i], &s, &c);
        on = (f & kActive) ? T(1) : T(0);
    };

    // ---- the lane's ROWS pixels, the halo rows above and below them, and (lane = side * ROWS + row) the halo columns of the wave ---------------------
    constexpr int CSL = OcLds<T>::csRows(ROWS, AP_LDS, LMV), CSR = ROWS - CSL;      // rows whose cos / sin live in LDS / in registers
    T* const myCs = reinterpret_cast<T*>(ocLds + OcLds<T>::base(ROWS, AP_LDS, LMV)) + tid;      // + ((row - CSR) * 2 + {0: cos, 1: sin}) * 512
    T p[ROWS][3], r[ROWS][3], cs[CSR > 0 ? CSR : 1][2];
    T dl[DELTA_GLB ? 1 : ROWS][3], ap[AP_LDS ? 1 : ROWS][3];
    unsigned fl[(ROWS + 3) / 4];
    T* const myB = bL + tid;      // LM: + (row * 3 + component) * 512
#pragma unroll
    for (int j = 0; j < (ROWS + 3) / 4; ++j) fl[j] = 0;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        unsigned f; T on;
        T cj, sj;
        loadPixel(x, yBase + j, p[j], r[j], cj, sj, on, f, true);
        if (j < CSR) { cs[j < CSR ? j : 0][0] = cj; cs[j < CSR ? j : 0][1] = sj; }
        else { myCs[((j - CSR) * 2 + 0) * kOcBlock] = cj; myCs[((j - CSR) * 2 + 1) * kOcBlock] = sj; }
        fl[j >> 2] |= f << (8 * (j & 3));
        if (!DELTA_GLB) { dl[DELTA_GLB ? 0 : j][0] = 0; dl[DELTA_GLB ? 0 : j][1] = 0; dl[DELTA_GLB ? 0 : j][2] = 0; }
        if (LMV) { myB[(j * 3 + 0) * kOcBlock] = r[j][0]; myB[(j * 3 + 1) * kOcBlock] = r[j][1]; myB[(j * 3 + 2) * kOcBlock] = r[j][2]; }      // b = r_0 (solver.t:657)
    }
    T tc, ts, ton, bc, bs, bon;
    unsigned fh;      // flag bytes of the halo pixels above (bits 0-7) and below (8-15)
    T* const myRowP = rowP + (wave * 2) * 3 * kWave + lane;      // + side * 3 * 64 + component * 64
    T* const myRowR = rowR + (wave * 2) * 3 * kWave + lane;
    T* const myRowD = rowD + (wave * 2) * 3 * kWave + lane;      // (LM)
    {
        unsigned ft, fb; T pp[3], rr[3];
        loadPixel(x, yBase - 1, pp, rr, tc, ts, ton, ft);
        myRowP[0] = pp[0]; myRowP[kWave] = pp[1]; myRowP[2 * kWave] = pp[2]; myRowR[0] = rr[0]; myRowR[kWave] = rr[1]; myRowR[2 * kWave] = rr[2];
        if (LMV) { myRowD[0] = 0; myRowD[kWave] = 0; myRowD[2 * kWave] = 0; }
        loadPixel(x, yBase + ROWS, pp, rr, bc, bs, bon, fb);
        myRowP[3 * kWave] = pp[0]; myRowP[4 * kWave] = pp[1]; myRowP[5 * kWave] = pp[2]; myRowR[3 * kWave] = rr[0]; myRowR[4 * kWave] = rr[1]; myRowR[5 * kWave] = rr[2];
        if (LMV) { myRowD[3 * kWave] = 0; myRowD[4 * kWave] = 0; myRowD[5 * kWave] = 0; }
        fh = ft | (fb << 8);
    }
    const bool haloLane = lane < 2 * ROWS;
    const int hSide = haloLane ? lane / ROWS : 0, hRow = haloLane ? lane % ROWS : 0;
    if (haloLane) {
        OcH4<T> p4, r4, c4; unsigned f;
        T pp[3], rr[3];
        loadPixel(hSide ? x0 + kWave : x0 - 1, yBase + hRow, pp, rr, c4.v[0], c4.v[1], c4.v[2], f);
        p4.v[0] = pp[0]; p4.v[1] = pp[1]; p4.v[2] = pp[2]; p4.v[3] = 0; r4.v[0] = rr[0]; r4.v[1] = rr[1]; r4.v[2] = rr[2]; r4.v[3] = 0;
        c4.v[3] = (T)f;      // (0 .. 255: exact in either precision)
        const int h = (wave * ROWS + hRow) * 2 + hSide;
        sideP[h] = p4; sideR[h] = r4; sideC[h] = c4;
        OcH4<T> z4; z4.v[0] = z4.v[1] = z4.v[2] = z4.v[3] = 0;
        sideA[h] = z4; stageA[h] = z4;      // A p of a halo pixel beyond the tile grid stays 0
        if (LMV) sideD[h] = z4;
    }
    __syncthreads();

    auto flagOf = [&](int j) -> unsigned { return (fl[j >> 2] >> (8 * (j & 3))) & 0xffu; };
    auto rowQ = [&](const T (&v)[ROWS][3], int j) {
        Q<T> q{};
        const unsigned f = flagOf(j);
        q.ox = v[j][0]; q.oy = v[j][1]; q.a = v[j][2]; q.c = j < CSR ? cs[j < CSR ? j : 0][0] : myCs[((j - CSR) * 2 + 0) * kOcBlock]; q.s = j < CSR ? cs[j < CSR ? j : 0][1] : myCs[((j - CSR) * 2 + 1) * kOcBlock];
        q.on = (f & kActive) ? T(1) : T(0); q.fw = (f & kFit) ? wf2 : T(0);
        return q;
    };
    auto haloQ = [&](const T (&h)[3], T c, T s, T on) { Q<T> q{}; q.ox = h[0]; q.oy = h[1]; q.a = h[2]; q.c = c; q.s = s; q.on = on; return q; };
    auto tabIndex = [&](unsigned f, int& io, int& ia) { const int cnt = (int)((f >> kCountShift) & 7u); io = cnt + ((f & kFit) ? 5 : 0); ia = 10 + cnt; };
    auto mOf = [&](unsigned f, T& mo, T& ma) { int io, ia; tabIndex(f, io, ia); mo = mTab[io]; ma = mTab[ia]; };
    const int sideSel = lane == kWave - 1 ? 1 : 0;       // lane 63 looks right, lane 0 (and, unused, everyone else) left
    // Per-lane base addresses, so that every row's access is base + a compile-time offset (the DS instructions' immediate): with the row inside the index
    // expression the compiler keeps one address register per row and array -- 32 of them, spilled and reloaded at L2 latency in every row of the stencil.
    const OcH4<T>* const mySideP = sideP + (wave * ROWS) * 2 + sideSel;
    const OcH4<T>* const mySideC = sideC + (wave * ROWS) * 2 + sideSel;
    const OcH4<T>* const mySideD = sideD + (wave * ROWS) * 2 + sideSel;      // (LM)
    T* const myAp = apL + tid;
    // where the A p of this lane's pixels goes if the lane is a wave edge (lane 0: to whoever holds the column as its right halo; lane 63: as its left halo):
    // the neighbouring wave's sideA, or -- at a tile edge -- this wave's stageA, from where one lane per halo pixel sends it to the neighbouring tile
    const bool edgeLane = lane == 0 || lane == kWave - 1;
    OcH4<T>* const edgeDst = lane == 0 ? (wx > 0 ? sideA + ((wave - 1) * ROWS) * 2 + 1 : stageA + (wave * ROWS) * 2 + 0)
                                       : (wx + 1 < kOcWavesX ? sideA + ((wave + 1) * ROWS) * 2 + 0 : stageA + (wave * ROWS) * 2 + 1);
    const int nGroups = (K.G + kOcGroup - 1) / kOcGroup;
    // a tile's neighbours: tiles of this grid, or -- first / last tile row of a slab -- the edge tiles of the rank above / below (words in the peer window)
    const bool upRemote = ty == 0 && K.links.edgeSendUp != nullptr, downRemote = ty + 1 == K.tilesY && K.links.edgeSendDown != nullptr;
    const bool hasUp = ty > 0 || upRemote, hasDown = ty + 1 < K.tilesY || downRemote, hasLeft = tx > 0, hasRight = tx + 1 < K.tilesX;
    // the halo column this lane looks after: handed over inside the workgroup, by another tile, or by nobody (the image ends)
    const bool hIntra = haloLane && (hSide == 0 ? wx > 0 : wx + 1 < kOcWavesX);
    const bool hInter = haloLane && !hIntra && (hSide == 0 ? hasLeft : hasRight);
    bool failed = false;

#if OC_PROFILE
    __shared__ long long ocProf[kOcWaves * 16];
    long long ocPrev = wall_clock64();
    if (tid < kOcWaves * 16) ocProf[tid] = 0;
    __syncthreads();
#endif
    int pix0 = yBase * K.W + x;      // index of the lane's first pixel (may lie outside the image: only used where the pixel exists)

    // ---- the stencil: A v on the lane's ROWS pixels (own values: registers; halo rows / columns: LDS), one row at a time; sink(j, centre, ox, oy, oa) consumes a row ----
    // One row per scheduling region (sched_barrier): left to itself the scheduler interleaves the unrolled rows until the live temporaries fill the
    // register budget and beyond.  The wave-edge halo of row j + 1 is requested before row j's arithmetic.
    auto stencil = [&](const T (&v)[ROWS][3], const T* myRowV, const OcH4<T>* mySideV, T (&aFirst)[3], T (&aLast)[3], auto&& sink) {
        const T pt[3] = {myRowV[0], myRowV[kWave], myRowV[2 * kWave]};
        Q<T> prevQ = haloQ(pt, tc, ts, ton), curQ = rowQ(v, 0);
        PairOut<T> vert;
        { T t0 = 0, t1 = 0, t2 = 0; vert = iw_pairFull<0, 1, true>(prevQ, curQ, t0, t1, t2); }      // the pair (row above, row 0) that row 0 inherits
        OcH4<T> spN = mySideV[0], scN = mySideC[0];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const OcH4<T> sp = spN, sc = scN;
            if (j + 1 < ROWS) { spN = mySideV[(j + 1) * 2]; scN = mySideC[(j + 1) * 2]; }
            T pb[3] = {0, 0, 0};
            if (j + 1 == ROWS) { pb[0] = myRowV[3 * kWave]; pb[1] = myRowV[4 * kWave]; pb[2] = myRowV[5 * kWave]; }
            const Q<T> nextQ = (j + 1 < ROWS) ? rowQ(v, j + 1 < ROWS ? j + 1 : j) : haloQ(pb, bc, bs, bon);
            T ax = 0, ay = 0, aa = 0;
            {
                Q<T> rq{};
                rq.ox = ocFromRight(sp.v[0], curQ.ox); rq.oy = ocFromRight(sp.v[1], curQ.oy); rq.a = ocFromRight(sp.v[2], curQ.a);
                rq.c = ocFromRight(sc.v[0], curQ.c); rq.s = ocFromRight(sc.v[1], curQ.s); rq.on = ocFromRight(sc.v[2], curQ.on);
                iw_pairQ<1, 0, true>(curQ, rq, ax, ay, aa);
            }
            {
                Q<T> lq{};
                lq.ox = ocFromLeft(sp.v[0], curQ.ox); lq.oy = ocFromLeft(sp.v[1], curQ.oy); lq.a = ocFromLeft(sp.v[2], curQ.a);
                lq.c = ocFromLeft(sc.v[0], curQ.c); lq.s = ocFromLeft(sc.v[1], curQ.s); lq.on = ocFromLeft(sc.v[2], curQ.on);
                iw_pairQ<-1, 0, true>(curQ, lq, ax, ay, aa);
            }
            const PairOut<T> vn = iw_pairFull<0, 1, true>(curQ, nextQ, ax, ay, aa);      // towards the next row: formed here, inherited there
            iw_pairInherited(vert, prevQ.on, ax, ay, aa);
            vert = vn;
            T ox = curQ.on * (w2 * ax + curQ.fw * curQ.ox), oy = curQ.on * (w2 * ay + curQ.fw * curQ.oy), oa = curQ.on * (w2 * aa);
            if (LMV) {      // + CtC v (o.t:2076-2082), the way iw_pcgIter2 adds it
                int io, ia; tabIndex(flagOf(j), io, ia);
                const T co = cTab[io], ca = cTab[ia];
                ox += co * curQ.ox; oy += co * curQ.oy; oa += ca * curQ.a;
            }
            // (values pinned here: pure arithmetic otherwise sinks out of its scheduling region -- all rows' DPP results then wait, live, for one block of arithmetic at the end)
            asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oa));
            sink(j, curQ, ox, oy, oa);      // (by reference: the sink may replace A v by what the halo holders are to receive instead)
            if (!AP_LDS && j == 0) { aFirst[0] = ox; aFirst[1] = oy; aFirst[2] = oa; }
            if (j == ROWS - 1) { aLast[0] = ox; aLast[1] = oy; aLast[2] = oa; }
            if (edgeLane) { OcH4<T> e; e.v[0] = ox; e.v[1] = oy; e.v[2] = oa; e.v[3] = 0; edgeDst[j * 2] = e; }
            prevQ = curQ; curQ = nextQ;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- hand the edge A v to whoever holds those pixels as halo: LDS inside the workgroup, tagged words between workgroups ----------------------------
    auto handOver = [&](T (&aFirst)[3], T (&aLast)[3], unsigned tag, oc_u64* boxPar, unsigned rtag, int rpar) {
        auto box = [&](int tile, int sd) { return boxPar + ((long)tile * 4 + sd) * K.S.stride; };      // sd: 0 from above, 1 from below, 2 from the left, 3 from the right
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (AP_LDS && wy == 0 && hasUp) { aFirst[0] = myAp[0]; aFirst[1] = myAp[kOcBlock]; aFirst[2] = myAp[2 * kOcBlock]; }      // (not held across the 16 rows)
        if (wy > 0) { if (!AP_LDS) { T* h = rowA + ((wave - kOcWavesX) * 2 + 1) * 3 * kWave; h[ln] = aFirst[0]; h[kWave + ln] = aFirst[1]; h[2 * kWave + ln] = aFirst[2]; } }
        else if (upRemote) { oc_u64* d = K.links.edgeSendUp + rpar * K.links.edgeParityStride + (long)tx * K.S.stride; ocSend<true>(d, wx * kWave + ln, aFirst[0], rtag); ocSend<true>(d, kOcTileW + wx * kWave + ln, aFirst[1], rtag); ocSend<true>(d, 2 * kOcTileW + wx * kWave + ln, aFirst[2], rtag); }
        else if (hasUp) { oc_u64* d = box(g - K.tilesX, 1); ocSend(d, wx * kWave + ln, aFirst[0], tag); ocSend(d, kOcTileW + wx * kWave + ln, aFirst[1], tag); ocSend(d, 2 * kOcTileW + wx * kWave + ln, aFirst[2], tag); }
        if (wy + 1 < kOcWavesY) { if (!AP_LDS) { T* h = rowA + ((wave + kOcWavesX) * 2 + 0) * 3 * kWave; h[ln] = aLast[0]; h[kWave + ln] = aLast[1]; h[2 * kWave + ln] = aLast[2]; } }
        else if (downRemote) { oc_u64* d = K.links.edgeSendDown + rpar * K.links.edgeParityStride + (long)tx * K.S.stride; ocSend<true>(d, wx * kWave + ln, aLast[0], rtag); ocSend<true>(d, kOcTileW + wx * kWave + ln, aLast[1], rtag); ocSend<true>(d, 2 * kOcTileW + wx * kWave + ln, aLast[2], rtag); }
        else if (hasDown) { oc_u64* d = box(g + K.tilesX, 0); ocSend(d, wx * kWave + ln, aLast[0], tag); ocSend(d, kOcTileW + wx * kWave + ln, aLast[1], tag); ocSend(d, 2 * kOcTileW + wx * kWave + ln, aLast[2], tag); }
        // a tile-edge wave's column leaves with one lane per pixel (the LDS operations of one wave execute in order: what lane 0 / 63 staged above is there)
        if (haloLane && ((hSide == 0 && wx == 0 && hasLeft) || (hSide == 1 && wx == kOcWavesX - 1 && hasRight))) {
            const OcH4<T> e = stageA[(wave * ROWS + hRow) * 2 + hSide];
            oc_u64* d = hSide == 0 ? box(g - 1, 3) : box(g + 1, 2);
            const int idx = (wy * ROWS + hRow) * 3;
            ocSend(d, idx, e.v[0], tag); ocSend(d, idx + 1, e.v[1], tag); ocSend(d, idx + 2, e.v[2], tag);
        }
    };

    // ---- the grid-wide sums of one phase (v4: this lane's partial sums; on return TOT holds the totals, TOT[6] the bad flag); the A v handed over inside the workgroup
    // is collected behind the first barrier, what other tiles handed over inside the wait: at / ab / as receive the A v of the halo pixels above / below the lane's
    // column and of the halo pixel this lane looks after.  k: the PCG iteration (the rank hop of row slabs numbers its mailbox slots by it).
    auto gridWait = [&](double (&v4)[NS], unsigned tag, int par, oc_u64* boxPar, int k, unsigned rtag, int rpar, T (&at)[3], T (&ab)[3], T (&as)[3]) {
        auto box = [&](int tile, int sd) { return boxPar + ((long)tile * 4 + sd) * K.S.stride; };
        // the FIRST phase's waits double as the co-residency check (OnchipArgs::firstTicks: every workgroup posts before it waits, so passing them proves the grid resident)
        const long long to = tag == K.tag0 ? K.firstTicks : K.timeoutTicks;
        int tq = tid;      // (opaque per iteration, like fl / pix0: the addresses below are recomputed, not kept in registers across the whole loop)
        asm volatile("" : "+v"(tq));
#pragma unroll
        for (int q = 0; q < NS; ++q) { v4[q] = ocWaveSum63(v4[q]); if (lane == kWave - 1) red[q * kOcWaves + wave] = v4[q]; }
        OC_MARK(2);      // wave sums
        __syncthreads();
        OC_MARK(3);      // barrier: the slowest wave's stencil
        {
            const int ln = tq & (kWave - 1);      // (shadowed below: same value)
            if (AP_LDS) {      // the owner's A p itself: the last row of the wave above (4 waves = 256 threads back), the first row of the wave below
                if (wy > 0) { const T* h = apL + ((ROWS - 1) * 3) * kOcBlock + (tq - kOcWavesX * kWave); at[0] = h[0]; at[1] = h[kOcBlock]; at[2] = h[2 * kOcBlock]; }
                if (wy + 1 < kOcWavesY) { const T* h = apL + (tq + kOcWavesX * kWave); ab[0] = h[0]; ab[1] = h[kOcBlock]; ab[2] = h[2 * kOcBlock]; }
            } else {
                if (wy > 0) { const T* h = rowA + (wave * 2 + 0) * 3 * kWave; at[0] = h[ln]; at[1] = h[kWave + ln]; at[2] = h[2 * kWave + ln]; }
                if (wy + 1 < kOcWavesY) { const T* h = rowA + (wave * 2 + 1) * 3 * kWave; ab[0] = h[ln]; ab[1] = h[kWave + ln]; ab[2] = h[2 * kWave + ln]; }
            }
            if (hIntra) { const OcH4<T> e = sideA[(wave * ROWS + hRow) * 2 + hSide]; as[0] = e.v[0]; as[1] = e.v[1]; as[2] = e.v[2]; }
        }
        oc_u64* const slotPar = K.S.slots + (long)par * K.G * NW;
        if (tq < NW) {
            double s = 0;
            for (int w = 0; w < kOcWaves; ++w) s += red[(tq >> 1) * kOcWaves + w];
            const oc_u64 b = (oc_u64)__double_as_longlong(s);
            ocStore(slotPar + (long)g * NW + tq, tag, (tq & 1) ? (unsigned)(b >> 32) : (unsigned)b);
        }
        const int ln = tq & (kWave - 1);
        const bool leader = !K.flat && (g % kOcGroup) == 0;
        // What other tiles handed over was posted before their sums and arrives before the totals can: it is collected FIRST, inside the wait for the sums
        // (a request costs a fabric round trip even when the words are there).  Only a group's first workgroup, on whose total 15 others wait, sums first.
        auto collectInbox = [&]() {
            if (wy == 0 && upRemote) ocRecv3<true>(K.links.edgeRecvUp + rpar * K.links.edgeParityStride + (long)tx * K.S.stride, wx * kWave + ln, kOcTileW + wx * kWave + ln, 2 * kOcTileW + wx * kWave + ln, rtag, bad, to, at);
            else if (wy == 0 && hasUp) ocRecv3(box(g, 0), wx * kWave + ln, kOcTileW + wx * kWave + ln, 2 * kOcTileW + wx * kWave + ln, tag, bad, to, at);
            if (wy == kOcWavesY - 1 && downRemote) ocRecv3<true>(K.links.edgeRecvDown + rpar * K.links.edgeParityStride + (long)tx * K.S.stride, wx * kWave + ln, kOcTileW + wx * kWave + ln, 2 * kOcTileW + wx * kWave + ln, rtag, bad, to, ab);
            else if (wy == kOcWavesY - 1 && hasDown) ocRecv3(box(g, 1), wx * kWave + ln, kOcTileW + wx * kWave + ln, 2 * kOcTileW + wx * kWave + ln, tag, bad, to, ab);
            if (hInter) { const int idx = (wy * ROWS + hRow) * 3; ocRecv3(box(g, hSide == 0 ? 2 : 3), idx, idx + 1, idx + 2, tag, bad, to, as); }
        };
        if (!leader) collectInbox();
        OC_MARK(4);      // inbox
        if (LMV || K.flat) {      // every workgroup reads every slot and forms the group totals itself (same order as the tree: same bits); a lane's (up to 4 / 5) requests are in flight together
            constexpr int kPer = (kOcMaxTiles * NW + kOcBlock - 1) / kOcBlock;
            oc_u64 w[kPer];
            const int nW = K.G * NW;
            auto fetch = [&]() {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < kPer; ++u) { const int i = tq + u * kOcBlock; w[u] = ocLoad(slotPar + (i < nW ? i : tq)); }
#pragma unroll
                for (int u = 0; u < kPer; ++u) ok = ok && (unsigned)(w[u] >> 32) == tag;
                return ok;
            };
            if (tq < nW && !fetch()) {
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                for (;;) {
                    __builtin_amdgcn_s_sleep(1);
                    if (fetch()) break;
                    if ((++spins & 31u) == 0) {
                        if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        if (wall_clock64() - t0 > to) { __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kPer; ++u) { const int i = tq + u * kOcBlock; if (i < nW) W1[i] = (unsigned)w[u]; }
            __syncthreads();
            if (tq < nGroups * NS) {
                const int q = tq % NS, grp = tq / NS, n = min(kOcGroup, K.G - grp * kOcGroup);
                double s = 0;
                for (int m = 0; m < n; ++m) s += ocJoin(W1[(grp * kOcGroup + m) * NW + 2 * q], W1[(grp * kOcGroup + m) * NW + 2 * q + 1]);
                GS[grp * NS + q] = s;
            }
            __syncthreads();
        } else if constexpr (!LMV) {
            oc_u64* const topPar = K.S.groupSlots + (long)par * nGroups * 8;
            if ((g % kOcGroup) == 0) {      // the group's first workgroup adds its group's slots and posts the total
                const int grp = g / kOcGroup, n = min(kOcGroup, K.G - grp * kOcGroup);
                if (tq < n * 8) W1[tq] = ocAwait(slotPar + (long)grp * kOcGroup * 8 + tq, tag, bad, to);
                __syncthreads();
                if (tq < 8) {
                    const int q = tq >> 1;
                    double s = 0;
                    for (int m = 0; m < n; ++m) s += ocJoin(W1[m * 8 + 2 * q], W1[m * 8 + 2 * q + 1]);
                    const oc_u64 b = (oc_u64)__double_as_longlong(s);
                    ocStore(topPar + (long)grp * 8 + tq, tag, (tq & 1) ? (unsigned)(b >> 32) : (unsigned)b);
                }
                collectInbox();
            }
            if (tq < nGroups * 8) W2[tq] = ocAwait(topPar + tq, tag, bad, to);
            __syncthreads();
            if (tq < nGroups * 4) { const int q = tq & 3, grp = tq >> 2; GS[grp * 4 + q] = ocJoin(W2[grp * 8 + 2 * q], W2[grp * 8 + 2 * q + 1]); }
            __syncthreads();
        }
        if (tq < NS) { double s = 0; for (int grp = 0; grp < nGroups; ++grp) s += GS[grp * NS + tq]; TOT[tq] = s; }
        if constexpr (!LMV) {
            if (K.links.mailMine) {      // row slabs: the rank hop (also with a single rank: a 1-rank slab job measures the hop without the xGMI flight) -- workgroup 0 posts this rank's totals to every rank's mailbox, everybody adds the ranks' totals in rank order
                __syncthreads();
                const unsigned seq = K.links.seq0 + (unsigned)k;
                const long slotOff = (long)(seq % (unsigned)K.links.slots) * K.links.slotStride;
                if (g == 0 && tq < 8 * K.links.world) {
                    const int t = tq >> 3, w = tq & 7;
                    const oc_u64 b = (oc_u64)__double_as_longlong(TOT[w >> 1]);
                    ocStore<true>(K.links.mailDst[t] + slotOff + w, seq, (w & 1) ? (unsigned)(b >> 32) : (unsigned)b);
                }
                if (tq < 8 * K.links.world) W2[tq] = ocAwait<true>(K.links.mailMine + slotOff + (long)(tq >> 3) * K.links.rankStride + (tq & 7), seq, bad, to);
                __syncthreads();
                if (tq < 4) { double s = 0; for (int rk = 0; rk < K.links.world; ++rk) s += ocJoin(W2[rk * 8 + 2 * tq], W2[rk * 8 + 2 * tq + 1]); TOT[tq] = s; }
            }
        }
        if (tq == 0) reinterpret_cast<int*>(TOT + 6)[0] = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        OC_MARK(5);      // grid-wide sum
    };

    // Levenberg-Marquardt loop state (uniform over the grid: every workgroup decides from the same totals)
    unsigned phase = 0;            // LM: phases passed so far (an iteration that ends with the split residual reset has two)
    double accQ = 0;               // this lane's part of Q of the iteration just applied, on its way to the next phase's sums
    bool qPending = false;         // ... and whether there is one
    T Q0 = 0;                      // fetchQ before the loop (solver.t:1050): delta = 0, so exactly 0
    for (int k = 0; k < K.L; ++k) {
        // Everything derived from the flag bytes and the pixel index (activity and fit multipliers, table addresses, row addresses, bounds predicates) is
        // invariant over the solve; hoisted out of this loop it would occupy ~100 registers of a budget of 256.  The empty asm makes the sources opaque
        // per iteration, so each use recomputes its two or three instructions.
#pragma unroll
        for (int j = 0; j < (ROWS + 3) / 4; ++j) asm volatile("" : "+v"(fl[j]));
        asm volatile("" : "+v"(pix0), "+v"(fh));
        const unsigned tag = K.tag0 + (LMV ? phase : (unsigned)k);
        const int par = (int)(tag & 1u);
        if (k == K.failAt && g == 0 && tid == 0) __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        oc_u64* const boxPar = K.S.inbox + (long)par * K.G * 4 * K.S.stride;

        // ---- PCGStep1: A p_k on the lane's pixels, with the four sums ----------------------------------------------------------------------------
        double accDen = 0, accNum = 0, acc2 = 0, acc3 = 0;
        T aFirst[3], aLast[3];      // A p of the wave's first and last row: what the waves above and below hold as halo
        stencil(p, myRowP, mySideP, aFirst, aLast, [&](int j, const Q<T>& curQ, T& ox, T& oy, T& oa) {
            T moT, maT; mOf(flagOf(j), moT, maT);
            if (AP_LDS) { myAp[(j * 3 + 0) * kOcBlock] = ox; myAp[(j * 3 + 1) * kOcBlock] = oy; myAp[(j * 3 + 2) * kOcBlock] = oa; }
            else { ap[AP_LDS ? 0 : j][0] = ox; ap[AP_LDS ? 0 : j][1] = oy; ap[AP_LDS ? 0 : j][2] = oa; }
            {   // the sums of iw_pcgIter2, term for term: p.Ap from float products, the three expansion sums from exact double products of M, r, A p
                const double mo = (double)moT, ma = (double)maT;
                accDen += (double)(curQ.ox * ox + curQ.oy * oy + curQ.a * oa);
                const double rx = (double)r[j][0], ry = (double)r[j][1], ra = (double)r[j][2], dx = (double)ox, dy = (double)oy, da = (double)oa;
                const double mrx = mo * rx, mry = mo * ry, mra = ma * ra;
                accNum += mrx * rx + mry * ry + mra * ra;
                acc2 += mrx * dx + mry * dy + mra * da;
                acc3 += (mo * dx) * dx + (mo * dy) * dy + (ma * da) * da;
            }
            asm volatile("" : "+v"(accDen), "+v"(accNum), "+v"(acc2), "+v"(acc3));
        });
        OC_MARK(0);      // stencil
        // words that cross ranks carry the communicator-wide sequence number of this iteration's rank hop (it never repeats over the life of the peer window,
        // whatever plans ran on the communicator before; the plan-local tag0 restarts with every plan)
        const unsigned rtag = K.links.seq0 + (unsigned)k;
        const int rpar = (int)(rtag & 1u);
        handOver(aFirst, aLast, tag, boxPar, rtag, rpar);
        OC_MARK(1);      // edge hand-over sends

        T at[3] = {0, 0, 0}, ab[3] = {0, 0, 0}, as[3] = {0, 0, 0};      // A p of the halo pixels above / below the lane's column, and of the halo pixel this lane looks after
        {
            double v4[NS];
            v4[0] = accNum; v4[1] = accDen; v4[2] = acc2; v4[3] = acc3;
            if constexpr (LMV) v4[4] = accQ;
            gridWait(v4, tag, par, boxPar, k, rtag, rpar, at, ab, as);
        }
        if constexpr (LMV) { ++phase; accQ = 0; }
        // With delta in memory (ROWS = 16) it is read in chunks of CH rows, two chunks ahead of the update: the first request goes out HERE, behind the wait for
        // the sums, and returns (lines this lane wrote one iteration ago, still in its XCD's L2) while the halo copies are updated.  (All 48 values requested
        // before the wait held 48 more registers over the sum, and every scratch reload in between waited for all of them: vmcnt counts in order.)
        constexpr int CH = ROWS < 4 ? ROWS : 4;
        auto rowExists = [&](int j) { return xin && pix0 + j * K.W < K.yEnd * K.W; };      // (x < W: then y < yEnd is the same as pixel index < yEnd * W)
        auto rowIndex = [&](int j) { return rowExists(j) ? pix0 + j * K.W : 0; };     // 0: a valid address whose value is not used
        constexpr int NCH = (ROWS + CH - 1) / CH;
        T dN[DELTA_GLB ? NCH : 1][CH][3];      // (fully unrolled below: every chunk has its own registers, live from its request to its use -- two chunks at a time)
        auto requestDelta = [&](int c) {
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                const int i = rowIndex(c * CH + jj);
                const V2<T> dv = ((const V2<T>*)K.delta)[i];
                dN[DELTA_GLB ? c : 0][jj][0] = dv.x; dN[DELTA_GLB ? c : 0][jj][1] = dv.y; dN[DELTA_GLB ? c : 0][jj][2] = (K.delta + 2 * N)[i];
            }
        };
        if (DELTA_GLB && k > 0) { requestDelta(0); if (NCH > 1) requestDelta(NCH > 1 ? 1 : 0); }
        __builtin_amdgcn_sched_barrier(0);
        OC_MARK(6);      // delta requests
        const double aNumD = TOT[0], aDenD = TOT[1], s2 = TOT[2], s3 = TOT[3];
        if (reinterpret_cast<const int*>(TOT + 6)[0]) { failed = true; break; }      // uniform over the workgroup: a wait timed out somewhere
        if (!LMV && K.trace && g == 0 && tid == 0) { K.trace[4 * k] = aNumD; K.trace[4 * k + 1] = aDenD; K.trace[4 * k + 2] = s2; K.trace[4 * k + 3] = s3; }
        if constexpr (LMV) {      // the q early-out of iteration k - 1 (solver.t:1093-1102): nothing of iteration k has been applied yet
            if (qPending) {
                const T Q1 = (T)TOT[4];
                const T zeta = T(k) * (Q1 - Q0) / Q1;
                if (zeta < K.qTolerance) { if (K.trace && g == 0 && tid == 0) { K.trace[1] = (double)zeta; K.trace[0] = (double)(k + 1); } break; }
                Q0 = Q1;
            }
        }
        // the scalars of iw_pcgIter2's prologue (solver.t:456-459, 544-547 guards; beta numerator by expansion, clamped like the direct sum it replaces)
        const T aNum = (T)aNumD, aDen = (T)aDenD;
        const T alpha = (aDen > T(0)) ? aNum / aDen : T(0);
        const double bNumD = fmax(aNumD - 2.0 * (double)alpha * s2 + (double)alpha * (double)alpha * s3, 0.0);
        T beta = (aNum > T(0)) ? (T)bNumD / aNum : T(0);
        const bool last = k + 1 == K.L;
        const bool reset = LMV && ((k + 1) % K.resetPeriod) == 0;      // this iteration ends with the split residual reset (solver.t:1077-1083)

        if constexpr (LMV) {
            if (reset) {
                // ---- PCGStep2_1stHalf (solver.t:491-503): delta += alpha p, on the lane's pixels and on its halo copies -------------------------------------------
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c) dl[DELTA_GLB ? 0 : j][c] = ocFma(alpha, p[j][c], dl[DELTA_GLB ? 0 : j][c]);
                if (last) break;      // only delta survives the last iteration (its r, z, p and Q are dead)
#pragma unroll
                for (int u = 0; u < 6; ++u) myRowD[u * kWave] = ocFma(alpha, myRowP[u * kWave], myRowD[u * kWave]);
                if (haloLane) {
                    const int h = (wave * ROWS + hRow) * 2 + hSide;
                    OcH4<T> d4 = sideD[h]; const OcH4<T> p4 = sideP[h];
                    d4.v[0] = ocFma(alpha, p4.v[0], d4.v[0]); d4.v[1] = ocFma(alpha, p4.v[1], d4.v[1]); d4.v[2] = ocFma(alpha, p4.v[2], d4.v[2]);
                    sideD[h] = d4;
                }
                // ---- computeAdelta + PCGStep2_2ndHalf (solver.t:566-571, 505-534): r = b - (J^T J + CtC) delta, with sum M r^2 and Q -- phase B -----------------------
                const unsigned tagB = K.tag0 + phase;
                const int parB = (int)(tagB & 1u);
                oc_u64* const boxParB = K.S.inbox + (long)parB * K.G * 4 * K.S.stride;
                double accB = 0, accQB = 0;
                T dFirst[3], dLast[3];
                stencil(dl, myRowD, mySideD, dFirst, dLast, [&](int j, const Q<T>&, T& ox, T& oy, T& oa) {
                    T moT, maT; mOf(flagOf(j), moT, maT);
                    const T b0 = myB[(j * 3 + 0) * kOcBlock], b1 = myB[(j * 3 + 1) * kOcBlock], b2 = myB[(j * 3 + 2) * kOcBlock];
                    const T r0 = b0 - ox, r1 = b1 - oy, r2 = b2 - oa;
                    r[j][0] = r0; r[j][1] = r1; r[j][2] = r2;
                    const double mo = (double)moT, ma = (double)maT, rx = (double)r0, ry = (double)r1, ra = (double)r2;
                    accB += (mo * rx) * rx + (mo * ry) * ry + (ma * ra) * ra;
                    accQB += (double)(T(0.5) * (dl[DELTA_GLB ? 0 : j][0] * (r0 + b0))) + (double)(T(0.5) * (dl[DELTA_GLB ? 0 : j][1] * (r1 + b1))) + (double)(T(0.5) * (dl[DELTA_GLB ? 0 : j][2] * (r2 + b2)));
                    asm volatile("" : "+v"(accB), "+v"(accQB));
                    ox = r0; oy = r1; oa = r2;      // the halo holders receive the new r itself (no copy of b with them)
                });
                handOver(dFirst, dLast, tagB, boxParB, 0u, 0);
                T dt[3] = {0, 0, 0}, db[3] = {0, 0, 0}, ds[3] = {0, 0, 0};      // the new r of the halo pixels
                {
                    double v4[NS];
                    v4[0] = accB; v4[1] = 0; v4[2] = 0; v4[3] = 0; v4[4] = accQB;
                    gridWait(v4, tagB, parB, boxParB, k, 0u, 0, dt, db, ds);
                }
                ++phase;
                if (reinterpret_cast<const int*>(TOT + 6)[0]) { failed = true; break; }
                {      // the q test of THIS iteration (the split step delivers Q directly)
                    const T Q1 = (T)TOT[4];
                    const T zeta = T(k + 1) * (Q1 - Q0) / Q1;
                    if (zeta < K.qTolerance) { if (K.trace && g == 0 && tid == 0) { K.trace[1] = (double)zeta; K.trace[0] = (double)(k + 2); } break; }
                    Q0 = Q1;
                }
                qPending = false;
                const T bNum = (T)TOT[0];
                beta = (aNum > T(0)) ? bNum / aNum : T(0);      // PCGStep3's guard (solver.t:544-547)
                // r (as received) and p = M r + beta p on the halo copies (the bits of their owners), then on the lane's pixels
                {
                    T mo, ma, hm[6];
                    mOf(fh & 0xffu, mo, ma); hm[0] = mo; hm[1] = mo; hm[2] = ma;
                    mOf((fh >> 8) & 0xffu, mo, ma); hm[3] = mo; hm[4] = mo; hm[5] = ma;
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        const T rr = u < 3 ? dt[u % 3] : db[u % 3];
                        myRowR[u * kWave] = rr; myRowP[u * kWave] = ocFma(beta, myRowP[u * kWave], hm[u] * rr);
                    }
                    if (haloLane) {
                        const int h = (wave * ROWS + hRow) * 2 + hSide;
                        OcH4<T> p4 = sideP[h], r4;
                        mOf((unsigned)sideC[h].v[3], mo, ma);
                        r4.v[0] = ds[0]; r4.v[1] = ds[1]; r4.v[2] = ds[2]; r4.v[3] = 0;
                        p4.v[0] = ocFma(beta, p4.v[0], mo * r4.v[0]); p4.v[1] = ocFma(beta, p4.v[1], mo * r4.v[1]); p4.v[2] = ocFma(beta, p4.v[2], ma * r4.v[2]);
                        sideP[h] = p4; sideR[h] = r4;
                    }
                }
#pragma unroll
                for (int j = 0; j < ROWS; ++j) {
                    T mo, ma; mOf(flagOf(j), mo, ma);
                    p[j][0] = ocFma(beta, p[j][0], mo * r[j][0]); p[j][1] = ocFma(beta, p[j][1], mo * r[j][1]); p[j][2] = ocFma(beta, p[j][2], ma * r[j][2]);
                    asm volatile("" : "+v"(p[j][0]), "+v"(p[j][1]), "+v"(p[j][2]));
                }
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
        }

        // ---- PCGStep2 + PCGStep3: delta += alpha p;  r -= alpha A p;  p = M r + beta p  (after the last iteration only delta survives) ----------------
        // The same three fused operations on the halo copies: the bits of the pixel's owner.  CH rows per scheduling region.
        if (!last) {
            T mo, ma, hr[6], hp[6], hm[6];      // the halo pixels above (0..2) and below (3..5) the lane's column: all reads first, then the arithmetic, then the writes
#pragma unroll
            for (int u = 0; u < 6; ++u) { hr[u] = myRowR[u * kWave]; hp[u] = myRowP[u * kWave]; }
            if (LMV) {      // the halo copies of delta (the split residual reset applies A to it)
#pragma unroll
                for (int u = 0; u < 6; ++u) myRowD[u * kWave] = ocFma(alpha, hp[u], myRowD[u * kWave]);
            }
            mOf(fh & 0xffu, mo, ma); hm[0] = mo; hm[1] = mo; hm[2] = ma;
            mOf((fh >> 8) & 0xffu, mo, ma); hm[3] = mo; hm[4] = mo; hm[5] = ma;
#pragma unroll
            for (int u = 0; u < 6; ++u) { hr[u] = ocFma(-alpha, u < 3 ? at[u % 3] : ab[u % 3], hr[u]); hp[u] = ocFma(beta, hp[u], hm[u] * hr[u]); }
#pragma unroll
            for (int u = 0; u < 6; ++u) { myRowR[u * kWave] = hr[u]; myRowP[u * kWave] = hp[u]; }
            if (haloLane) {
                const int h = (wave * ROWS + hRow) * 2 + hSide;
                OcH4<T> p4 = sideP[h], r4 = sideR[h];
                mOf((unsigned)sideC[h].v[3], mo, ma);
                if (LMV) { OcH4<T> d4 = sideD[h]; d4.v[0] = ocFma(alpha, p4.v[0], d4.v[0]); d4.v[1] = ocFma(alpha, p4.v[1], d4.v[1]); d4.v[2] = ocFma(alpha, p4.v[2], d4.v[2]); sideD[h] = d4; }
                r4.v[0] = ocFma(-alpha, as[0], r4.v[0]); r4.v[1] = ocFma(-alpha, as[1], r4.v[1]); r4.v[2] = ocFma(-alpha, as[2], r4.v[2]);
                p4.v[0] = ocFma(beta, p4.v[0], mo * r4.v[0]); p4.v[1] = ocFma(beta, p4.v[1], mo * r4.v[1]); p4.v[2] = ocFma(beta, p4.v[2], ma * r4.v[2]);
                sideP[h] = p4; sideR[h] = r4;
            }
        }
        OC_MARK(7);      // halo update (mark 7)
#pragma unroll
        for (int c0 = 0; c0 < ROWS; c0 += CH) {
            T dC[CH][3];
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                if (DELTA_GLB) { const int c = DELTA_GLB ? c0 / CH : 0; dC[jj][0] = (k == 0) ? T(0) : dN[c][jj][0]; dC[jj][1] = (k == 0) ? T(0) : dN[c][jj][1]; dC[jj][2] = (k == 0) ? T(0) : dN[c][jj][2]; }
                else { dC[jj][0] = dl[DELTA_GLB ? 0 : c0 + jj][0]; dC[jj][1] = dl[DELTA_GLB ? 0 : c0 + jj][1]; dC[jj][2] = dl[DELTA_GLB ? 0 : c0 + jj][2]; }
            }
            if (DELTA_GLB && k > 0 && c0 / CH + 2 < NCH) requestDelta(c0 / CH + 2 < NCH ? c0 / CH + 2 : 0);
            T aC[CH][3];
#pragma unroll
            for (int jj = 0; jj < CH; ++jj)
#pragma unroll
                for (int c = 0; c < 3; ++c) aC[jj][c] = AP_LDS ? myAp[((c0 + jj) * 3 + c) * kOcBlock] : ap[AP_LDS ? 0 : c0 + jj][c];
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                const int j = c0 + jj;
                T mo, ma; mOf(flagOf(j), mo, ma);
                const T m[3] = {mo, mo, ma};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    dC[jj][c] = ocFma(alpha, p[j][c], dC[jj][c]);
                    if (!last) {
                        r[j][c] = ocFma(-alpha, aC[jj][c], r[j][c]);
                        if (LMV) accQ += (double)(T(0.5) * (dC[jj][c] * (r[j][c] + myB[(j * 3 + c) * kOcBlock])));      // Q = 1/2 sum delta . (r + b), solver.t:483-485
                        p[j][c] = ocFma(beta, p[j][c], m[c] * r[j][c]);
                    }
                }
                if (DELTA_GLB) {
                    if (rowExists(j)) { const int i = pix0 + j * K.W; ((V2<T>*)K.delta)[i] = V2<T>{dC[jj][0], dC[jj][1]}; (K.delta + 2 * N)[i] = dC[jj][2]; }
                } else { dl[DELTA_GLB ? 0 : j][0] = dC[jj][0]; dl[DELTA_GLB ? 0 : j][1] = dC[jj][1]; dl[DELTA_GLB ? 0 : j][2] = dC[jj][2]; }
                asm volatile("" : "+v"(p[j][0]), "+v"(p[j][1]), "+v"(p[j][2]), "+v"(r[j][0]), "+v"(r[j][1]), "+v"(r[j][2]));
                if (!DELTA_GLB) asm volatile("" : "+v"(dl[DELTA_GLB ? 0 : j][0]), "+v"(dl[DELTA_GLB ? 0 : j][1]), "+v"(dl[DELTA_GLB ? 0 : j][2]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LMV) qPending = true;
        OC_MARK(8);      // own update
    }
#if OC_PROFILE
    __syncthreads();
    if (tid < kOcWaves * 16 && K.prof) K.prof[(long)g * kOcWaves * 16 + tid] = ocProf[tid];
#endif
    if (!DELTA_GLB && !failed) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int y = yBase + j;
            if (xin && y < K.yEnd) { const long i = (long)y * K.W + x; ((V2<T>*)K.delta)[i] = V2<T>{dl[DELTA_GLB ? 0 : j][0], dl[DELTA_GLB ? 0 : j][1]}; K.delta[2 * N + i] = dl[DELTA_GLB ? 0 : j][2]; }
        }
    }
}

// Behind an on-chip Levenberg-Marquardt solve (whose update the solver applies itself: savePreviousUnknowns + PCGLinearUpdate): tell the host if a wait timed out.
__global__ void iw_relayBad(const int* __restrict__ bad, int* hostErr) {
    if (threadIdx.x == 0 && __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) __hip_atomic_store(hostErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// PCGLinearUpdate X += delta (solver.t:552-557) behind the on-chip solve -- unless one of its waits timed out: then the unknowns stay untouched, the host is
// told (pinned word) and redoes the linear solve with the streaming kernels.  Row slabs: `verdict` is the all-reduced count of ranks whose kernel failed, so
// either every rank applies its delta or none does.
template <class T>
__global__ __launch_bounds__(kBlock) void iw_applyDelta(T* __restrict__ XO, T* __restrict__ XA, const T* __restrict__ delta, long N, const int* __restrict__ bad,
                                                        const double* __restrict__ verdict, int* hostErr, int* stepErr = nullptr) {
    const bool fail = verdict ? verdict[0] != 0.0 : __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (fail) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            __hip_atomic_store(hostErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (stepErr) __hip_atomic_store(stepErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (EnergyOps::onChipStepSlot: which of several enqueued steps this was)
        }
        return;
    }
    V2<T>* xO = (V2<T>*)XO; const V2<T>* dO = (const V2<T>*)delta;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const V2<T> xv = xO[i], dv = dO[i];
        xO[i] = V2<T>{xv.x + dv.x, xv.y + dv.y};
        XA[i] = XA[i] + delta[2 * N + i];
    }
}
// a rank's own verdict as the double the communicator all-reduces (row slabs)
__global__ void iw_badToScalar(const int* __restrict__ bad, int force, double* out) {
    if (threadIdx.x == 0) out[0] = (force || __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1.0 : 0.0;
}

}  // namespace
}  // namespace optamd
