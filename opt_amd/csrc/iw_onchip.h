// image_warping, Gauss-Newton, unit lattice: the WHOLE PCG linear solve as one persistent launch whose loop state never leaves the chip.
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
// and synchronises the grid twice per iteration with 8-byte {payload, tag} words (one relaxed agent-scope store each, no fences, no cache write-backs;
// MI355X_MICROARCH.md "handoff-1to1" / "allgather"; measured in tools/microbench_gridsync.hip):
//   halo   a workgroup's tile is 256 x 2 ROWS pixels (8 waves: 4 across, 2 down).  Wave-edge rows and columns of p_k travel through LDS inside a workgroup
//          and through per-tile inboxes in global memory between workgroups;
//   sum    the four sums of the iteration (alphaDen = p.Ap, alphaNum = sum M r^2, s2 = sum M r.Ap, s3 = sum M Ap^2; beta by expansion as in iw_pcgIter2,
//          energy.h PcgIterArgs) as a two-level tree: 16 workgroups per group, group totals posted by the group's first workgroup, every workgroup adds the
//          group totals in group order -- the same bits everywhere, so alpha and beta agree on the whole grid without a broadcast.
// Every wait is bounded by the device's wall clock; a time-out raises K.S.bad, every workgroup leaves the loop at its next sum, nothing is applied to the
// unknowns (iw_applyDelta checks the flag) and the host falls back to the streaming loop.  The grid must be co-resident (one workgroup per CU): the launcher
// checks tiles <= CUs x occupancy.
#pragma once
#include "iw_device.h"

namespace optamd {
namespace {

typedef unsigned long long oc_u64;
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
template <class T>
struct OnchipArgs {
    int W, H, tilesX, tilesY, G;
    const T* r0; const T* p0;           // solver layout: [O.x O.y] x N, then [a] x N
    const T* Angle; const uint8_t* flags;
    T* delta;                           // out: sum alpha_k p_k
    T w_fit, w_reg;
    int L; unsigned tag0;               // iterations; tag of iteration 0 (tags never repeat over the life of the buffers)
    int flat;                           // 1: every workgroup reads every workgroup's slot (small grids); 0: two-level tree
    OnchipSync S;
    double* trace;                      // [L][4] = alphaNum, alphaDen, s2, s3 of every iteration (written by workgroup 0), or nullptr
    long long timeoutTicks;
    long long* prof;                    // OC_PROFILE builds: [G][8] ticks per phase, else nullptr
    int failAt;                         // test hook (OPT_AMD_ONCHIP_FAIL_AT): workgroup 0 raises `bad` in this iteration as a timed-out wait would; -1: never
};

__device__ __forceinline__ oc_u64 ocLoad(const oc_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ocStore(oc_u64* p, unsigned tag, unsigned half) { __hip_atomic_store(p, ((oc_u64)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Waits until *src carries `tag`; returns the payload.  Bounded: after timeoutTicks of the 100 MHz wall clock -- or as soon as another waiter has given up --
// the wait falls through with whatever is there (the caller's loop ends at its next sum).
__device__ __forceinline__ unsigned ocAwait(const oc_u64* src, unsigned tag, int* bad, long long timeoutTicks) {
    oc_u64 v = ocLoad(src);
    if ((unsigned)(v >> 32) != tag) {
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            __builtin_amdgcn_s_sleep(1);
            v = ocLoad(src);
            if ((unsigned)(v >> 32) == tag) break;
            if ((++spins & 31u) == 0) {
                if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if (wall_clock64() - t0 > timeoutTicks) { __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
    }
    return (unsigned)v;
}
// one scalar of the halo as tagged words: a float is one word, a double two
__device__ __forceinline__ void ocSend(oc_u64* box, int idx, float v, unsigned tag) { ocStore(box + idx, tag, __float_as_uint(v)); }
__device__ __forceinline__ void ocSend(oc_u64* box, int idx, double v, unsigned tag) {
    const oc_u64 b = (oc_u64)__double_as_longlong(v);
    ocStore(box + 2 * idx, tag, (unsigned)b); ocStore(box + 2 * idx + 1, tag, (unsigned)(b >> 32));
}
__device__ __forceinline__ void ocRecv(const oc_u64* box, int idx, unsigned tag, int* bad, long long to, float& v) { v = __uint_as_float(ocAwait(box + idx, tag, bad, to)); }
__device__ __forceinline__ void ocRecv(const oc_u64* box, int idx, unsigned tag, int* bad, long long to, double& v) {
    const unsigned lo = ocAwait(box + 2 * idx, tag, bad, to), hi = ocAwait(box + 2 * idx + 1, tag, bad, to);
    v = __longlong_as_double((long long)(((oc_u64)hi << 32) | lo));
}
__device__ __forceinline__ double ocJoin(unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((oc_u64)hi << 32) | lo)); }

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

template <class T> struct __attribute__((aligned(16))) OcH4 { T v[4]; };      // {ox, oy, a, -} or {cos, sin, on, -} of one halo pixel

// LDS carve-up (bytes), shared by the kernel and the launcher
template <class T> struct OcLds {
    static constexpr size_t ap(int rows, bool apLds) { return apLds ? (size_t)rows * 3 * kOcBlock * sizeof(T) : 0; }
    static constexpr size_t rowHalo() { return (size_t)kOcWaves * 2 * 3 * kWave * sizeof(T); }
    static constexpr size_t side(int rows) { return (size_t)kOcWaves * rows * 2 * sizeof(OcH4<T>); }
    static constexpr size_t stage(int rows) { return ((size_t)kOcWaves * rows * 3 * sizeof(T) + 15) / 16 * 16; }
    static constexpr size_t tail() { return (4 * kOcWaves + kOcGroup * 4 + 8) * sizeof(double) + (kOcMaxTiles * 8 + kOcGroup * 8) * sizeof(unsigned) + 16 * sizeof(T) + 16; }
    static constexpr size_t total(int rows, bool apLds) { return ap(rows, apLds) + rowHalo() + 2 * side(rows) + stage(rows) + tail(); }
};

// Development builds (opt_amd/build.py build_variant with OC_PROFILE=1; tools/onchip_bench.py --profile): thread 0 of every workgroup accumulates the wall-clock
// ticks (100 MHz) it spends in each phase of an iteration and leaves them in K.prof[workgroup][8].
#ifndef OC_PROFILE
#define OC_PROFILE 0
#endif
#if OC_PROFILE
#define OC_MARK(i) do { if (tid == 0) { const long long t_ = wall_clock64(); ocProf[i] += t_ - ocPrev; ocPrev = t_; } } while (0)
#else
#define OC_MARK(i) do { } while (0)
#endif

template <class T, int ROWS, bool AP_LDS, bool DELTA_GLB>
__global__ __launch_bounds__(kOcBlock, 2) void iw_onchipPcg(OnchipArgs<T> K) {
    static_assert(3 * ROWS <= kWave, "a wave hands its edge column over with one lane per scalar");
    extern __shared__ __attribute__((aligned(16))) unsigned char ocLds[];
    T* apL = reinterpret_cast<T*>(ocLds);                                                       // [ROWS * 3][512]: conflict-free [row][component][thread]
    T* rowHalo = reinterpret_cast<T*>(ocLds + OcLds<T>::ap(ROWS, AP_LDS));                      // [wave][0 = from above, 1 = from below][3][64]
    OcH4<T>* sideP = reinterpret_cast<OcH4<T>*>(reinterpret_cast<unsigned char*>(rowHalo) + OcLds<T>::rowHalo());      // [wave][row][0 = from the left, 1 = from the right]
    OcH4<T>* sideC = sideP + kOcWaves * ROWS * 2;                                               // the same pixels' cos, sin, on (constant over the solve)
    T* stage = reinterpret_cast<T*>(sideC + kOcWaves * ROWS * 2);                               // [wave][ROWS * 3]: the edge column a tile-edge wave sends out
    double* red = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(stage) + OcLds<T>::stage(ROWS));          // [4][waves]
    double* GS = red + 4 * kOcWaves;                                                            // [groups][4]
    double* TOT = GS + kOcGroup * 4;                                                            // [4] + the bad flag
    unsigned* W1 = reinterpret_cast<unsigned*>(TOT + 8);                                        // [<= 256 workgroups][8]
    unsigned* W2 = W1 + kOcMaxTiles * 8;                                                        // [<= 16 groups][8]
    T* mTab = reinterpret_cast<T*>(W2 + kOcGroup * 8);                                          // guardedInvert(diag J^T J) by flag byte, as in iw_pcgIter2 (PRE == 3)

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6, wx = wave & (kOcWavesX - 1), wy = wave / kOcWavesX;
    const int g = blockIdx.x, tx = g % K.tilesX, ty = g / K.tilesX;
    const int x0 = tx * kOcTileW + wx * kWave, x = x0 + lane, yBase = (ty * kOcWavesY + wy) * ROWS;
    const long N = (long)K.W * K.H;
    const bool xin = x < K.W;
    const T w2 = K.w_reg * K.w_reg, wf2 = K.w_fit * K.w_fit;
    int* const bad = K.S.bad;
    const long long to = K.timeoutTicks;

    if (tid < 15) {      // the table of iw_pcgIter2: same accumulation order as iw_evalJTF, so the entries are the solver's preconditioner values bit for bit
        const int t = tid, cnt = t < 10 ? t % 5 : t - 10;
        const T w = K.w_reg;
        T d = 0;
        if (t < 10) { for (int n = 0; n < cnt; ++n) d += w * w + w * w; if (t >= 5) d += K.w_fit * K.w_fit; }
        else for (int n = 0; n < cnt; ++n) d += (w * T(1)) * (w * T(1));
        const T sq = T(1) + sqrt(d);
        mTab[t] = T(1) / (sq * sq);
    }
    // cos, sin, activity of a pixel that may lie outside the image (then: inactive)
    auto pixelConst = [&](int xx, int yy, T& c, T& s, T& on) {
        const bool ok = xx >= 0 && xx < K.W && yy >= 0 && yy < K.H;
        const long i = ok ? (long)yy * K.W + xx : 0;
        const int f = K.flags[i];
        sincosT(K.Angle[i], &s, &c);
        on = (ok && (f & kActive)) ? T(1) : T(0);
    };

    // ---- the lane's ROWS pixels ---------------------------------------------------------------------------------------------------------------
    T p[ROWS][3], r[ROWS][3], cs[ROWS][2];
    T dl[DELTA_GLB ? 1 : ROWS][3], ap[AP_LDS ? 1 : ROWS][3];
    unsigned fl[(ROWS + 3) / 4];
#pragma unroll
    for (int j = 0; j < (ROWS + 3) / 4; ++j) fl[j] = 0;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int y = yBase + j;
        const bool ok = xin && y < K.H;
        const long i = ok ? (long)y * K.W + x : 0;
        const unsigned f = ok ? (unsigned)K.flags[i] : 0u;
        const V2<T> po = ((const V2<T>*)K.p0)[i], ro = ((const V2<T>*)K.r0)[i];
        const T pa = K.p0[2 * N + i], ra = K.r0[2 * N + i];
        p[j][0] = ok ? po.x : T(0); p[j][1] = ok ? po.y : T(0); p[j][2] = ok ? pa : T(0);
        r[j][0] = ok ? ro.x : T(0); r[j][1] = ok ? ro.y : T(0); r[j][2] = ok ? ra : T(0);
        sincosT(K.Angle[i], &cs[j][1], &cs[j][0]);
        fl[j >> 2] |= f << (8 * (j & 3));
        if (!DELTA_GLB) { dl[DELTA_GLB ? 0 : j][0] = 0; dl[DELTA_GLB ? 0 : j][1] = 0; dl[DELTA_GLB ? 0 : j][2] = 0; }
    }
    // the rows above and below the wave's rows at this lane's column, and (lane = side * ROWS + row) the columns left and right of the wave
    T tc, ts, ton, bc, bs, bon;
    pixelConst(x, yBase - 1, tc, ts, ton);
    pixelConst(x, yBase + ROWS, bc, bs, bon);
    if (lane < 2 * ROWS) {
        const int sd = lane / ROWS, row = lane % ROWS;
        OcH4<T> c4; c4.v[3] = 0;
        pixelConst(sd ? x0 + kWave : x0 - 1, yBase + row, c4.v[0], c4.v[1], c4.v[2]);
        sideC[(wave * ROWS + row) * 2 + sd] = c4;
        OcH4<T> z4; z4.v[0] = z4.v[1] = z4.v[2] = z4.v[3] = 0;
        sideP[(wave * ROWS + row) * 2 + sd] = z4;         // stays 0 where the image (or the tile grid) ends
    }
    __syncthreads();

    auto flagOf = [&](int j) -> unsigned { return (fl[j >> 2] >> (8 * (j & 3))) & 0xffu; };
    auto rowQ = [&](int j) {
        Q<T> q{};
        const unsigned f = flagOf(j);
        q.ox = p[j][0]; q.oy = p[j][1]; q.a = p[j][2]; q.c = cs[j][0]; q.s = cs[j][1];
        q.on = (f & kActive) ? T(1) : T(0); q.fw = (f & kFit) ? wf2 : T(0);
        return q;
    };
    auto haloQ = [&](const T (&h)[3], T c, T s, T on) { Q<T> q{}; q.ox = h[0]; q.oy = h[1]; q.a = h[2]; q.c = c; q.s = s; q.on = on; return q; };
    const int sideSel = lane == kWave - 1 ? 1 : 0;       // lane 63 looks right, lane 0 (and, unused, everyone else) left
    // Per-lane base addresses, so that every row's access is base + a compile-time offset (the DS instructions' immediate): with the row inside the index
    // expression the compiler keeps one address register per row and array -- 32 of them, spilled and reloaded at L2 latency in every row of the stencil.
    const OcH4<T>* const mySideP = sideP + (wave * ROWS) * 2 + sideSel;
    const OcH4<T>* const mySideC = sideC + (wave * ROWS) * 2 + sideSel;
    T* const myAp = apL + tid;
    const int nGroups = (K.G + kOcGroup - 1) / kOcGroup;
    const bool hasUp = ty > 0, hasDown = ty + 1 < K.tilesY, hasLeft = tx > 0, hasRight = tx + 1 < K.tilesX;
    bool failed = false;

#if OC_PROFILE
    __shared__ long long ocProf[8];
    long long ocPrev = wall_clock64();
    if (tid < 8) ocProf[tid] = 0;
    __syncthreads();
#endif
    int pix0 = yBase * K.W + x;      // index of the lane's first pixel (may lie outside the image: only used where the pixel exists)
    for (int k = 0; k < K.L; ++k) {
        // Everything derived from the flag bytes and the pixel index (activity and fit multipliers, table addresses, row addresses, bounds predicates) is
        // invariant over the solve; hoisted out of this loop it would occupy ~100 registers of a budget of 256.  The empty asm makes the sources opaque
        // per iteration, so each use recomputes its two or three instructions.
#pragma unroll
        for (int j = 0; j < (ROWS + 3) / 4; ++j) asm volatile("" : "+v"(fl[j]));
        asm volatile("" : "+v"(pix0));
        const unsigned tag = K.tag0 + (unsigned)k;
        const int par = (int)(tag & 1u);
        if (k == K.failAt && g == 0 && tid == 0) __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        oc_u64* const boxPar = K.S.inbox + (long)par * K.G * 4 * K.S.stride;
        auto box = [&](int tile, int sd) { return boxPar + ((long)tile * 4 + sd) * K.S.stride; };      // sd: 0 from above, 1 from below, 2 from the left, 3 from the right

        int ln = lane;
        asm volatile("" : "+v"(ln));
        // ---- hand the edges of p_k to the neighbours: LDS inside the workgroup, tagged words between workgroups --------------------------------
        if (wy > 0) { T* h = rowHalo + ((wave - kOcWavesX) * 2 + 1) * 3 * kWave; h[ln] = p[0][0]; h[kWave + ln] = p[0][1]; h[2 * kWave + ln] = p[0][2]; }
        else if (hasUp) { oc_u64* d = box(g - K.tilesX, 1); ocSend(d, wx * kWave + ln, p[0][0], tag); ocSend(d, kOcTileW + wx * kWave + ln, p[0][1], tag); ocSend(d, 2 * kOcTileW + wx * kWave + ln, p[0][2], tag); }
        if (wy + 1 < kOcWavesY) { T* h = rowHalo + ((wave + kOcWavesX) * 2 + 0) * 3 * kWave; h[ln] = p[ROWS - 1][0]; h[kWave + ln] = p[ROWS - 1][1]; h[2 * kWave + ln] = p[ROWS - 1][2]; }
        else if (hasDown) { oc_u64* d = box(g + K.tilesX, 0); ocSend(d, wx * kWave + ln, p[ROWS - 1][0], tag); ocSend(d, kOcTileW + wx * kWave + ln, p[ROWS - 1][1], tag); ocSend(d, 2 * kOcTileW + wx * kWave + ln, p[ROWS - 1][2], tag); }
        if (ln == 0) {
            if (wx > 0) {
#pragma unroll
                for (int j = 0; j < ROWS; ++j) { OcH4<T> v; v.v[0] = p[j][0]; v.v[1] = p[j][1]; v.v[2] = p[j][2]; v.v[3] = 0; (sideP + ((wave - 1) * ROWS) * 2 + 1)[j * 2] = v; }
            } else if (hasLeft) {
#pragma unroll
                for (int j = 0; j < ROWS; ++j) { T* st = stage + wave * ROWS * 3; st[j * 3 + 0] = p[j][0]; st[j * 3 + 1] = p[j][1]; st[j * 3 + 2] = p[j][2]; }
            }
        }
        if (ln == kWave - 1) {
            if (wx + 1 < kOcWavesX) {
#pragma unroll
                for (int j = 0; j < ROWS; ++j) { OcH4<T> v; v.v[0] = p[j][0]; v.v[1] = p[j][1]; v.v[2] = p[j][2]; v.v[3] = 0; (sideP + ((wave + 1) * ROWS) * 2)[j * 2] = v; }
            } else if (hasRight) {
#pragma unroll
                for (int j = 0; j < ROWS; ++j) { T* st = stage + wave * ROWS * 3; st[j * 3 + 0] = p[j][0]; st[j * 3 + 1] = p[j][1]; st[j * 3 + 2] = p[j][2]; }
            }
        }
        // a tile-edge wave's column leaves with one ln per scalar (the LDS operations of one wave execute in order: the staged values are there)
        if (wx == 0 && hasLeft && ln < 3 * ROWS) ocSend(box(g - 1, 3), wy * ROWS * 3 + ln, stage[wave * ROWS * 3 + ln], tag);
        if (wx == kOcWavesX - 1 && hasRight && ln < 3 * ROWS) ocSend(box(g + 1, 2), wy * ROWS * 3 + ln, stage[wave * ROWS * 3 + ln], tag);
        __syncthreads();
        OC_MARK(0);

        // ---- collect this wave's halo ----------------------------------------------------------------------------------------------------------
        T ht[3] = {0, 0, 0}, hb[3] = {0, 0, 0};
        if (wy > 0) { const T* h = rowHalo + (wave * 2 + 0) * 3 * kWave; ht[0] = h[ln]; ht[1] = h[kWave + ln]; ht[2] = h[2 * kWave + ln]; }
        else if (hasUp) { const oc_u64* s = box(g, 0); ocRecv(s, wx * kWave + ln, tag, bad, to, ht[0]); ocRecv(s, kOcTileW + wx * kWave + ln, tag, bad, to, ht[1]); ocRecv(s, 2 * kOcTileW + wx * kWave + ln, tag, bad, to, ht[2]); }
        if (wy + 1 < kOcWavesY) { const T* h = rowHalo + (wave * 2 + 1) * 3 * kWave; hb[0] = h[ln]; hb[1] = h[kWave + ln]; hb[2] = h[2 * kWave + ln]; }
        else if (hasDown) { const oc_u64* s = box(g, 1); ocRecv(s, wx * kWave + ln, tag, bad, to, hb[0]); ocRecv(s, kOcTileW + wx * kWave + ln, tag, bad, to, hb[1]); ocRecv(s, 2 * kOcTileW + wx * kWave + ln, tag, bad, to, hb[2]); }
        if (wx == 0 && hasLeft && ln < 3 * ROWS) { T v; ocRecv(box(g, 2), wy * ROWS * 3 + ln, tag, bad, to, v); sideP[(wave * ROWS + ln / 3) * 2 + 0].v[ln % 3] = v; }
        if (wx == kOcWavesX - 1 && hasRight && ln < 3 * ROWS) { T v; ocRecv(box(g, 3), wy * ROWS * 3 + ln, tag, bad, to, v); sideP[(wave * ROWS + ln / 3) * 2 + 1].v[ln % 3] = v; }

        OC_MARK(1);
        // ---- PCGStep1: A p_k on the lane's pixels, with the four sums ----------------------------------------------------------------------------
        // One row per scheduling region (sched_barrier): left to itself the scheduler interleaves the unrolled rows until the live temporaries fill the
        // register budget and beyond.  The wave-edge halo of row j + 1 is requested before row j's arithmetic.
        double accDen = 0, accNum = 0, acc2 = 0, acc3 = 0;
        {
            Q<T> prevQ = haloQ(ht, tc, ts, ton), curQ = rowQ(0);
            PairOut<T> vert;
            { T t0 = 0, t1 = 0, t2 = 0; vert = iw_pairFull<0, 1, true>(prevQ, curQ, t0, t1, t2); }      // the pair (row above, row 0) that row 0 inherits
            OcH4<T> spN = mySideP[0], scN = mySideC[0];
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const OcH4<T> sp = spN, sc = scN;
                if (j + 1 < ROWS) { spN = mySideP[(j + 1) * 2]; scN = mySideC[(j + 1) * 2]; }
                const unsigned f = flagOf(j);
                const int cnt = (int)((f >> kCountShift) & 7u), io = cnt + ((f & kFit) ? 5 : 0);
                const T moT = mTab[io], maT = mTab[10 + cnt];
                const Q<T> nextQ = (j + 1 < ROWS) ? rowQ(j + 1 < ROWS ? j + 1 : j) : haloQ(hb, bc, bs, bon);
                T ax = 0, ay = 0, aa = 0;
                {
                    Q<T> rt{};
                    rt.ox = ocFromRight(sp.v[0], curQ.ox); rt.oy = ocFromRight(sp.v[1], curQ.oy); rt.a = ocFromRight(sp.v[2], curQ.a);
                    rt.c = ocFromRight(sc.v[0], curQ.c); rt.s = ocFromRight(sc.v[1], curQ.s); rt.on = ocFromRight(sc.v[2], curQ.on);
                    iw_pairQ<1, 0, true>(curQ, rt, ax, ay, aa);
                }
                {
                    Q<T> lf{};
                    lf.ox = ocFromLeft(sp.v[0], curQ.ox); lf.oy = ocFromLeft(sp.v[1], curQ.oy); lf.a = ocFromLeft(sp.v[2], curQ.a);
                    lf.c = ocFromLeft(sc.v[0], curQ.c); lf.s = ocFromLeft(sc.v[1], curQ.s); lf.on = ocFromLeft(sc.v[2], curQ.on);
                    iw_pairQ<-1, 0, true>(curQ, lf, ax, ay, aa);
                }
                const PairOut<T> vn = iw_pairFull<0, 1, true>(curQ, nextQ, ax, ay, aa);      // towards the next row: formed here, inherited there
                iw_pairInherited(vert, prevQ.on, ax, ay, aa);
                vert = vn;
                T ox = curQ.on * (w2 * ax + curQ.fw * curQ.ox), oy = curQ.on * (w2 * ay + curQ.fw * curQ.oy), oa = curQ.on * (w2 * aa);
                // (values pinned here: pure arithmetic otherwise sinks out of its scheduling region -- all rows' DPP results then wait, live, for one block of arithmetic at the end)
                asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oa));
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
                prevQ = curQ; curQ = nextQ;
                asm volatile("" : "+v"(accDen), "+v"(accNum), "+v"(acc2), "+v"(acc3));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the delta of the first rows is requested before the wait for the sums: it does not depend on them
        constexpr int CH = ROWS < 4 ? ROWS : 4;      // rows per chunk of the update below
        // pixel index of the lane's row j, or 0 (a valid address whose value is not used) where the pixel does not exist
        auto rowExists = [&](int j) { return xin && pix0 + j * K.W < (int)N; };      // (x < W: then y < H is the same as pixel index < N)
        auto rowIndex = [&](int j) { return rowExists(j) ? pix0 + j * K.W : 0; };
        // With delta in memory (ROWS = 16), ALL of the lane's delta is requested here, before the wait for the sums: the stencil's temporaries are dead, so the
        // 3 ROWS registers are free, and the reads (L2-resident lines this lane wrote one iteration ago) complete while the grid-wide sum is in flight.
        T dG[DELTA_GLB ? ROWS : 1][3];
        if (DELTA_GLB && k > 0) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const int i = rowIndex(j);
                const V2<T> dv = ((const V2<T>*)K.delta)[i];
                dG[DELTA_GLB ? j : 0][0] = dv.x; dG[DELTA_GLB ? j : 0][1] = dv.y; dG[DELTA_GLB ? j : 0][2] = (K.delta + 2 * N)[i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        OC_MARK(2);

        // ---- the grid-wide sums -------------------------------------------------------------------------------------------------------------------
        {
            int tq = tid;      // (opaque per iteration, like fl / pix0 above: the addresses below are recomputed, not kept in registers across the whole loop)
            asm volatile("" : "+v"(tq));
            double v4[4] = {accNum, accDen, acc2, acc3};
#pragma unroll
            for (int q = 0; q < 4; ++q) { v4[q] = waveReduceSum(v4[q]); if (lane == 0) red[q * kOcWaves + wave] = v4[q]; }
            __syncthreads();
            oc_u64* const slotPar = K.S.slots + (long)par * K.G * 8;
            if (tq < 8) {
                double s = 0;
                for (int w = 0; w < kOcWaves; ++w) s += red[(tq >> 1) * kOcWaves + w];
                const oc_u64 b = (oc_u64)__double_as_longlong(s);
                ocStore(slotPar + (long)g * 8 + tq, tag, (tq & 1) ? (unsigned)(b >> 32) : (unsigned)b);
            }
            OC_MARK(3);
            if (K.flat) {      // small grids: every workgroup reads every slot and forms the group totals itself (same order as the tree: same bits)
                for (int i = tq; i < K.G * 8; i += kOcBlock) W1[i] = ocAwait(slotPar + i, tag, bad, to);
                __syncthreads();
                if (tq < nGroups * 4) {
                    const int q = tq & 3, grp = tq >> 2, n = min(kOcGroup, K.G - grp * kOcGroup);
                    double s = 0;
                    for (int m = 0; m < n; ++m) s += ocJoin(W1[(grp * kOcGroup + m) * 8 + 2 * q], W1[(grp * kOcGroup + m) * 8 + 2 * q + 1]);
                    GS[grp * 4 + q] = s;
                }
                __syncthreads();
            } else {
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
                }
                if (tq < nGroups * 8) W2[tq] = ocAwait(topPar + tq, tag, bad, to);
                __syncthreads();
                if (tq < nGroups * 4) { const int q = tq & 3, grp = tq >> 2; GS[grp * 4 + q] = ocJoin(W2[grp * 8 + 2 * q], W2[grp * 8 + 2 * q + 1]); }
                __syncthreads();
            }
            if (tq < 4) { double s = 0; for (int grp = 0; grp < nGroups; ++grp) s += GS[grp * 4 + tq]; TOT[tq] = s; }
            if (tq == 0) reinterpret_cast<int*>(TOT + 4)[0] = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
        OC_MARK(4);
        const double aNumD = TOT[0], aDenD = TOT[1], s2 = TOT[2], s3 = TOT[3];
        if (reinterpret_cast<const int*>(TOT + 4)[0]) { failed = true; break; }      // uniform over the workgroup: a wait timed out somewhere
        if (K.trace && g == 0 && tid == 0) { K.trace[4 * k] = aNumD; K.trace[4 * k + 1] = aDenD; K.trace[4 * k + 2] = s2; K.trace[4 * k + 3] = s3; }
        // the scalars of iw_pcgIter2's prologue (solver.t:456-459, 544-547 guards; beta numerator by expansion, clamped like the direct sum it replaces)
        const T aNum = (T)aNumD, aDen = (T)aDenD;
        const T alpha = (aDen > T(0)) ? aNum / aDen : T(0);
        const double bNumD = fmax(aNumD - 2.0 * (double)alpha * s2 + (double)alpha * (double)alpha * s3, 0.0);
        const T beta = (aNum > T(0)) ? (T)bNumD / aNum : T(0);
        const bool last = k + 1 == K.L;

        // ---- PCGStep2 + PCGStep3: delta += alpha p;  r -= alpha A p;  p = M r + beta p  (after the last iteration only delta survives) ----------------
        // CH rows per scheduling region; with delta in memory the next chunk's delta is in flight while this one is updated.
#pragma unroll
        for (int c0 = 0; c0 < ROWS; c0 += CH) {
            T dC[CH][3];
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                if (DELTA_GLB) { dC[jj][0] = (k == 0) ? T(0) : dG[DELTA_GLB ? c0 + jj : 0][0]; dC[jj][1] = (k == 0) ? T(0) : dG[DELTA_GLB ? c0 + jj : 0][1]; dC[jj][2] = (k == 0) ? T(0) : dG[DELTA_GLB ? c0 + jj : 0][2]; }
                else { dC[jj][0] = dl[DELTA_GLB ? 0 : c0 + jj][0]; dC[jj][1] = dl[DELTA_GLB ? 0 : c0 + jj][1]; dC[jj][2] = dl[DELTA_GLB ? 0 : c0 + jj][2]; }
            }
            T aC[CH][3];
#pragma unroll
            for (int jj = 0; jj < CH; ++jj)
#pragma unroll
                for (int c = 0; c < 3; ++c) aC[jj][c] = AP_LDS ? myAp[((c0 + jj) * 3 + c) * kOcBlock] : ap[AP_LDS ? 0 : c0 + jj][c];
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                const int j = c0 + jj;
                const unsigned f = flagOf(j);
                const int cnt = (int)((f >> kCountShift) & 7u), io = cnt + ((f & kFit) ? 5 : 0);
                const T m[3] = {mTab[io], mTab[io], mTab[10 + cnt]};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    dC[jj][c] = dC[jj][c] + alpha * p[j][c];
                    if (!last) {
                        r[j][c] = r[j][c] - alpha * aC[jj][c];
                        const T z = m[c] * r[j][c];
                        p[j][c] = z + beta * p[j][c];
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
        OC_MARK(5);
    }
#if OC_PROFILE
    __syncthreads();
    if (tid < 8 && K.prof) K.prof[(long)g * 8 + tid] = ocProf[tid];
#endif
    if (!DELTA_GLB && !failed) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int y = yBase + j;
            if (xin && y < K.H) { const long i = (long)y * K.W + x; ((V2<T>*)K.delta)[i] = V2<T>{dl[DELTA_GLB ? 0 : j][0], dl[DELTA_GLB ? 0 : j][1]}; K.delta[2 * N + i] = dl[DELTA_GLB ? 0 : j][2]; }
        }
    }
}

// PCGLinearUpdate X += delta (solver.t:552-557) behind the on-chip solve -- unless one of its waits timed out: then the unknowns stay untouched, the host is
// told (pinned word) and redoes the linear solve with the streaming kernels.
template <class T>
__global__ __launch_bounds__(kBlock) void iw_applyDelta(T* __restrict__ XO, T* __restrict__ XA, const T* __restrict__ delta, long N, const int* __restrict__ bad, int* hostErr) {
    if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(hostErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    V2<T>* xO = (V2<T>*)XO; const V2<T>* dO = (const V2<T>*)delta;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const V2<T> xv = xO[i], dv = dO[i];
        xO[i] = V2<T>{xv.x + dv.x, xv.y + dv.y};
        XA[i] = XA[i] + delta[2 * N + i];
    }
}

}  // namespace
}  // namespace optamd
