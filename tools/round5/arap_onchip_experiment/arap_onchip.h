// arap_mesh_deformation, Gauss-Newton, symmetric graph, float: the WHOLE PCG linear solve as one persistent launch (BASELINE config 4: 500 k vertices, GN 20 x 100).
//
// Included by energy_graph.hip (host side: ArapOps::pcgSolveOnChip).  What it replaces: the reference's loop `for lIter = 0, lIterations do PCGStep1; PCGStep1_Graph;
// PCGStep2; PCGStep3 end` (solverGPUGaussNewton.t:1056-1092 with the graph kernels :687-706) -- four launches and three same-address-atomic sums per iteration
// there, two launches here until round 4 (arap_flatStepRec + arap_applySym: 57 us per iteration at 500 k vertices, both passes over a cache-resident working set at
// half of HBM's peak rate).  500 k vertices x (p, r, A p) x 6 scalars fit the register files of the chip, so:
//   * a thread OWNS up to VPT vertices for the whole solve: p, r and the A p of the current iteration live in its registers; delta is read-modified-written in the
//     solver's delta vector (lines only this thread touches), M, the slots {U_v - U_u, u} and the sines / cosines come from memory every iteration (constant: cached);
//   * a workgroup owns a CHUNK of consecutive vertices and keeps their search directions in LDS, where most of a vertex's neighbours are found (a mesh in vertex
//     order: ~3/4 of them); what other chunks need -- the search directions of the chunk's boundary vertices -- is PUBLISHED once per iteration as three 16-byte
//     pieces of two {payload, tag} words each (onchip_sync.h: written through with sc1, read with sc1, no fence, no cache write-back), and every workgroup fetches
//     its halo list (the remote vertices its own read, found once per graph: ao_buildHalo / ao_buildIndex) into LDS in ONE bulk round trip, polling until the tags
//     carry the iteration.  (A first version that polled each remote neighbour where the gather needed it took 66 us per iteration: sixteen dependent round trips
//     per thread.)  The tags double as the point-to-point synchronisation between PCGStep3 of iteration k and the gather of iteration k + 1: there is ONE
//     grid-wide wait per iteration, the one for the sums (alphaDen = sum |J p|^2 + fit terms, alphaNum = sum M r^2, s2 = sum M r.Ap, s3 = sum M Ap^2; beta by
//     expansion exactly as arap_applySym / arap_flatStepRec form it, energy.h PcgIterArgs);
//   * the sums travel like iw_onchipPcg's: every workgroup posts four doubles as eight tagged words, every workgroup reads all of them and adds in workgroup order --
//     the same alpha and beta everywhere, no broadcast;
//   * workgroups are dealt to the XCDs in contiguous eighths of the vertex range (blockIdx % 8 is the XCD), as arap_applySym does.
// The arithmetic per half-edge pair is arap_applySym's, term for term (same association), with one lane per vertex walking its whole out-list.  Every wait is bounded by
// the wall clock; a time-out raises `bad`, every workgroup leaves at its next sum, nothing is applied and the host redoes the step with the two-kernel loop.
#pragma once
#include "../../../opt_amd/csrc/onchip_sync.h"

namespace optamd {
namespace {

constexpr int kAoBlock = 512, kAoWaves = kAoBlock / kWave, kAoMaxGrid = 256;
constexpr int kAoHaloCap = 1024, kAoHaloPerThread = kAoHaloCap / kAoBlock;      // remote vertices a workgroup may depend on (chunks of 2048 vertices in Morton order of the rest positions: a few hundred on a mesh)
typedef unsigned int ao_u4 __attribute__((ext_vector_type(4)));

struct ArapOnchipSync { oc_u64* slots; int* bad; int* hostErr; };      // slots: [2][G][8]
template <class T>
struct ArapOnchipArgs {
    ArapArgs<T> A;
    const int* outOff; const ArapSlot<T>* slots; const ArapRec<T>* rec;      // rec: only the sines / cosines (floats 6 .. 11 of a record) are read
    const void* aoSlots;                                                      // AoSlot<T> per half-edge slot (ao_buildSlots)
    const int* haloList; const int* haloCount; const unsigned char* boundary; // [chunks][kAoHaloCap] remote vertices a chunk reads; their number; per vertex: some other chunk reads it
    const int* perm;                                                          // position -> vertex: the vertices in Morton order of their rest positions (a chunk = kAoBlock * VPT consecutive positions)
    const T* r0; const T* p0; const T* diag;                                  // solver layout: [O.xyz] x N, then [a.xyz] x N; diag = raw diag(J^T J): M = guardedInvert(diag) (k_initFinish)
    long long* prof;                                                          // [G][8] ticks per phase (development), or nullptr
    T* delta;                                                                 // in: 0 (PCGInit1); out: sum alpha_k p_k
    T* XO; T* XA;                                                             // the unknowns: X += delta at the end (PCGLinearUpdate, solver.t:552-557)
    ao_u4* gran;                                                              // [2][N][3]: a boundary vertex's published search direction, {p.x, tag, p.y, tag} {p.z, tag, pa.x, tag} {pa.y, tag, pa.z, tag}
    int L, G; unsigned tag0;
    ArapOnchipSync S;
    double* trace;                                                            // [L][4] = alphaNum, alphaDen, s2, s3 (workgroup 0), or nullptr
    long long timeoutTicks; int failAt;
};

template <class T> __device__ __forceinline__ T aoGi(T x) { const T sq = T(1) + sqrt(x); return T(1) / (sq * sq); }      // guardedInvert (solver.t:323-332)
__device__ __forceinline__ ao_u4 aoLoad(__amdgpu_buffer_rsrc_t rs, unsigned byteOff) { return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byteOff, 0, /*aux: sc1*/ 16); }
__device__ __forceinline__ void aoStore(__amdgpu_buffer_rsrc_t rs, unsigned byteOff, float a, float b, unsigned tag) {
    __builtin_amdgcn_raw_buffer_store_b128(ao_u4{__float_as_uint(a), tag, __float_as_uint(b), tag}, rs, (int)byteOff, 0, /*aux: sc1*/ 16);
}
// which chunk of NV consecutive vertices workgroup g owns: workgroup g runs on XCD g % 8 and takes the (g / 8)-th chunk of that XCD's contiguous eighth of the chunks
__host__ __device__ inline long aoChunkOf(int g, int G) { return (long)(g % 8) * (G / 8) + g / 8; }

// ---- once per graph: the vertices in Morton order of their rest positions, so that a chunk of consecutive positions is a compact patch of the mesh (a chunk of a mesh
// in its file's vertex order is a strip of whole rows: 70 % of its vertices then have a neighbour in another chunk, and the exchange costs 18-28 us per iteration) ----
__device__ __forceinline__ unsigned aoOrdered(float x) { const unsigned f = __float_as_uint(x); return f ^ ((f >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float aoUnordered(unsigned o) { return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu)); }
template <class T>
__global__ __launch_bounds__(kBlock) void ao_bbox(long N, const T* __restrict__ U, unsigned* __restrict__ mm) {      // mm[0..2] = min, mm[3..5] = max (ordered encoding)
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0, 0, 0};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x)
        for (int c = 0; c < 3; ++c) { const unsigned o = aoOrdered((float)U[3 * i + c]); lo[c] = min(lo[c], o); hi[c] = max(hi[c], o); }
    for (int c = 0; c < 3; ++c) { atomicMin(mm + c, lo[c]); atomicMax(mm + 3 + c, hi[c]); }
}
__device__ __forceinline__ unsigned aoSpread10(unsigned v) { v &= 1023u; v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u; return v; }
template <class T>
__global__ __launch_bounds__(kBlock) void ao_mortonKeys(long N, const T* __restrict__ U, const unsigned* __restrict__ mm, unsigned* __restrict__ keys, int* __restrict__ ids) {
    float lo[3], sc[3];
    float ext = 0.f;      // ONE scale for the three axes (the largest extent): a nearly flat mesh must not spend Morton bits on its thickness
    for (int c = 0; c < 3; ++c) { lo[c] = aoUnordered(mm[c]); ext = fmaxf(ext, aoUnordered(mm[3 + c]) - lo[c]); }
    for (int c = 0; c < 3; ++c) sc[c] = ext > 0.f ? 1023.f / ext : 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        unsigned q[3];
        for (int c = 0; c < 3; ++c) q[c] = (unsigned)fminf(1023.f, fmaxf(0.f, ((float)U[3 * i + c] - lo[c]) * sc[c]));
        keys[i] = aoSpread10(q[0]) | (aoSpread10(q[1]) << 1) | (aoSpread10(q[2]) << 2);
        ids[i] = (int)i;
    }
}
__global__ __launch_bounds__(kBlock) void ao_invert(long N, const int* __restrict__ perm, int* __restrict__ inv) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) inv[perm[i]] = (int)i;
}

// ---- once per graph: who reads whom across chunks ---------------------------------------------------------------------------------------------------------------
// Pass 1, per vertex u: for every chunk other than its own that holds a neighbour of u, u joins that chunk's halo list (once); posIn[k] = u's position in the halo
// list of the chunk of slot k's neighbour (or -1 if that neighbour is in u's own chunk).  The order inside a list is whatever the atomics give: it only decides where a
// value sits in LDS.
template <class T>
__global__ __launch_bounds__(kBlock) void ao_buildHalo(long N, int NV, const int* __restrict__ inv, const int* __restrict__ outOff, const ArapSlot<T>* __restrict__ slots, int* __restrict__ haloCount,
                                                       int* __restrict__ haloList, int* __restrict__ posIn, unsigned char* __restrict__ boundary) {
    for (long u = blockIdx.x * (long)blockDim.x + threadIdx.x; u < N; u += (long)gridDim.x * blockDim.x) {
        const int cu = inv[u] / NV, bo = outOff[u], eo = outOff[u + 1];
        bool any = false;
        for (int k = bo; k < eo; ++k) {
            const int cv = inv[slots[k].nbr] / NV;
            int pos = -1;
            if (cv != cu) {
                any = true;
                for (int k2 = bo; k2 < k && pos < 0; ++k2) if (inv[slots[k2].nbr] / NV == cv) pos = posIn[k2];      // (this thread wrote it a moment ago)
                if (pos < 0) { pos = atomicAdd(haloCount + cv, 1); if (pos < kAoHaloCap) haloList[(long)cv * kAoHaloCap + pos] = (int)u; }
            }
            posIn[k] = pos;
        }
        boundary[u] = any ? 1 : 0;
    }
}
// Pass 2, per slot (v -> u): the LDS index of u's search direction in v's workgroup: u - first vertex of the chunk, or NV + u's position in the chunk's halo list
// (found through the reverse slot (u -> v): the graph is symmetric).
template <class T>
__global__ __launch_bounds__(kBlock) void ao_buildIndex(long N, int NV, const int* __restrict__ inv, const int* __restrict__ outOff, const ArapSlot<T>* __restrict__ slots, const int* __restrict__ posIn, int* __restrict__ sidx) {
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < N; v += (long)gridDim.x * blockDim.x) {
        const int cv = inv[v] / NV;
        for (int k = outOff[v]; k < outOff[v + 1]; ++k) {
            const int u = slots[k].nbr;
            int idx;
            if (inv[u] / NV == cv) idx = inv[u] - cv * NV;
            else {
                idx = -1;
                for (int k2 = outOff[u]; k2 < outOff[u + 1] && idx < 0; ++k2) if (slots[k2].nbr == (int)v) idx = NV + posIn[k2];
            }
            sidx[k] = idx;
        }
    }
}

// Once per Gauss-Newton step: everything the gather needs from a half-edge (v -> u) that does not change during the linear solve, as three 16-byte pieces --
// {U_v - U_u, LDS index of u's search direction} and the derivative columns E_k = dR3/da_k(a_u) (U_u - U_v) of the REVERSE edge (arap_cols of the neighbour's
// sines / cosines: the neighbour's record then need not be visited, which was a second dependent round trip to L2 per batch, nor its 21 coefficients formed).
template <class T> struct alignas(16) AoSlot { T ux, uy, uz; int li; T e0x, e0y, e0z, e1x, e1y, e1z, e2x, e2y; };      // (E_2.z = 0)
template <class T>
__global__ __launch_bounds__(kBlock) void ao_buildSlots(long nE, const ArapSlot<T>* __restrict__ slots, const int* __restrict__ sidx, const ArapRec<T>* __restrict__ rec, AoSlot<T>* __restrict__ out) {
    for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < nE; k += (long)gridDim.x * blockDim.x) {
        const ArapSlot<T> sl = slots[k];
        const ArapRec<T> nb = rec[sl.nbr];
        const ArapCoef<T> cu = arap_coef(nb.sa, nb.ca, nb.sb, nb.cb, nb.sg, nb.cg);
        const V3<T> un{-sl.ux, -sl.uy, -sl.uz};
        V3<T> E0, E1, E2;
        arap_cols(cu, un, E0, E1, E2);
        out[k] = AoSlot<T>{sl.ux, sl.uy, sl.uz, sidx[k], E0.x, E0.y, E0.z, E1.x, E1.y, E1.z, E2.x, E2.y};
    }
}

template <class T, int VPT>
__global__ __launch_bounds__(kAoBlock, 2) void arap_onchipPcg(ArapOnchipArgs<T> K) {
    static_assert(sizeof(T) == 4, "one 8-byte word carries a 4-byte payload and its tag");
    constexpr int NV = VPT * kAoBlock, NL = NV + kAoHaloCap;      // own vertices, then the halo
    extern __shared__ __attribute__((aligned(16))) unsigned char aoLds[];
    T* const pL = reinterpret_cast<T*>(aoLds);      // [6][NL]: the search direction of the workgroup's own vertices and of the remote vertices they read
    __shared__ double red[4 * kAoWaves], GS[4];
    __shared__ unsigned W1[kAoMaxGrid * 8];
    __shared__ int badL;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6, g = blockIdx.x;
    const long N = K.A.N, offA = 3 * N;
    const long chunk = aoChunkOf(g, K.G);
    const long base = chunk * (long)NV;
    int* const bad = K.S.bad;
    const long long to = K.timeoutTicks;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)K.gran, 0, -1, 0x00020000);
    const T w = K.A.w_reg;

    T p[VPT][6], r[VPT][6], ap[VPT][6], wf2[VPT];
    bool ok[VPT], pub[VPT];
    int vid[VPT];      // the vertices this thread owns (positions base + j * 512 + tid of the Morton order)
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const long pos = base + (long)j * kAoBlock + tid;
        ok[j] = pos < N;
        vid[j] = ok[j] ? K.perm[pos] : 0;
        const long iv = vid[j];
        const V3<T> po = ld3(K.p0, iv), pa = ld3(K.p0 + offA, iv), ro = ld3(K.r0, iv), ra = ld3(K.r0 + offA, iv);
        p[j][0] = po.x; p[j][1] = po.y; p[j][2] = po.z; p[j][3] = pa.x; p[j][4] = pa.y; p[j][5] = pa.z;
        r[j][0] = ro.x; r[j][1] = ro.y; r[j][2] = ro.z; r[j][3] = ra.x; r[j][4] = ra.y; r[j][5] = ra.z;
        const bool valid = K.A.Constraints[3 * iv] >= T(-999999.9);      // arap_mesh_deformation.t:13
        wf2[j] = (ok[j] && valid) ? K.A.w_fit * K.A.w_fit : T(0);
        pub[j] = ok[j] && K.boundary[iv] != 0;
    }
    // the halo entries this thread fetches every iteration
    const int nHalo = min(K.haloCount[chunk], kAoHaloCap);
    int hid[kAoHaloPerThread];
#pragma unroll
    for (int e = 0; e < kAoHaloPerThread; ++e) { const int h = tid + e * kAoBlock; hid[e] = h < nHalo ? K.haloList[chunk * kAoHaloCap + h] : -1; }

    // own search directions into LDS, the boundary ones published; then the remote ones fetched (one bulk round trip: every request of a thread in flight together)
    auto shareAndPublish = [&](unsigned tag) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int lj = j * kAoBlock + tid;
#pragma unroll
            for (int c = 0; c < 6; ++c) pL[c * NL + lj] = p[j][c];
            if (pub[j]) {
                const long i = vid[j];
                const unsigned off = (unsigned)((((long)(tag & 1u) * N + i) * 3) * 16);
                aoStore(gr, off, p[j][0], p[j][1], tag); aoStore(gr, off + 16, p[j][2], p[j][3], tag); aoStore(gr, off + 32, p[j][4], p[j][5], tag);
            }
        }
    };
    auto fetchHalo = [&](unsigned tag) {
        const long parBase = (long)(tag & 1u) * N;
        ao_u4 q0[kAoHaloPerThread], q1[kAoHaloPerThread], q2[kAoHaloPerThread];
        auto fetch = [&]() {
            bool ready = true;
#pragma unroll
            for (int e = 0; e < kAoHaloPerThread; ++e) {
                const unsigned off = (unsigned)(((parBase + max(hid[e], 0)) * 3) * 16);
                q0[e] = aoLoad(gr, off); q1[e] = aoLoad(gr, off + 16); q2[e] = aoLoad(gr, off + 32);
            }
#pragma unroll
            for (int e = 0; e < kAoHaloPerThread; ++e)
                ready = ready && (hid[e] < 0 || (q0[e].y == tag && q0[e].w == tag && q1[e].y == tag && q1[e].w == tag && q2[e].y == tag && q2[e].w == tag));
            return ready;
        };
        if (hid[0] >= 0 && !fetch()) {      // (entries are dealt in order: a thread without a first entry has none)
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
        for (int e = 0; e < kAoHaloPerThread; ++e) {
            if (hid[e] < 0) continue;
            const int lh = NV + tid + e * kAoBlock;
            pL[0 * NL + lh] = __uint_as_float(q0[e].x); pL[1 * NL + lh] = __uint_as_float(q0[e].z); pL[2 * NL + lh] = __uint_as_float(q1[e].x);
            pL[3 * NL + lh] = __uint_as_float(q1[e].z); pL[4 * NL + lh] = __uint_as_float(q2[e].x); pL[5 * NL + lh] = __uint_as_float(q2[e].z);
        }
    };
    shareAndPublish(K.tag0);
    fetchHalo(K.tag0);
    __syncthreads();

    bool failed = false;
    long long tPh[6] = {0, 0, 0, 0, 0, 0}, tPrev = wall_clock64();
#define AO_MARK(i) do { if (K.prof && tid == 0) { const long long t_ = wall_clock64(); tPh[i] += t_ - tPrev; tPrev = t_; } } while (0)
    for (int k = 0; k < K.L; ++k) {
        const unsigned tag = K.tag0 + (unsigned)k;
        if (k == K.failAt && g == 0 && tid == 0) __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- PCGStep1 + PCGStep1_Graph: A p on the thread's vertices; every neighbour's search direction comes from LDS --------------------------------------------
        double accDen = 0, accNum = 0, acc2 = 0, acc3 = 0;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const long i = vid[j];
            const long iv = i;
            const ao_u4* const mine = reinterpret_cast<const ao_u4*>(K.rec + iv);
            const ao_u4 t1 = mine[1], t2 = mine[2];      // {ay, az, sa, ca} {sb, cb, sg, cg}
            const ArapCoef<T> cv = arap_coef(__uint_as_float(t1.z), __uint_as_float(t1.w), __uint_as_float(t2.x), __uint_as_float(t2.y), __uint_as_float(t2.z), __uint_as_float(t2.w));
            const int bo = K.outOff[iv], eo = ok[j] ? K.outOff[iv + 1] : bo;
            const V3<T> pv{p[j][0], p[j][1], p[j][2]}, pav{p[j][3], p[j][4], p[j][5]};
            T s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
            constexpr int B = 3;      // neighbours requested together: one round trip to L2 / the Infinity Cache per batch (three independent 16-byte pieces per slot)
            const AoSlot<T>* const SL = reinterpret_cast<const AoSlot<T>*>(K.aoSlots);
            for (int k0 = bo; k0 < eo; k0 += B) {
                AoSlot<T> sl[B]; T wm[B];
#pragma unroll
                for (int b = 0; b < B; ++b) { const int kk = k0 + b; wm[b] = kk < eo ? w : T(0); sl[b] = SL[min(kk, eo - 1)]; }
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const T wj = wm[b];
                    const int li = sl[b].li;
                    const T npx = pL[0 * NL + li], npy = pL[1 * NL + li], npz = pL[2 * NL + li];
                    const T nax = pL[3 * NL + li], nay = pL[4 * NL + li], naz = pL[5 * NL + li];
                    const V3<T> u{sl[b].ux, sl[b].uy, sl[b].uz};
                    V3<T> D0, D1, D2;
                    arap_cols(cv, u, D0, D1, D2);
                    {   // out-edge (v -> u): J p and D_k . J p  (arap_applySym / arap_edges<3> with v0 = v)
                        const T jx = w * (pv.x - npx) - w * (D0.x * pav.x + D1.x * pav.y + D2.x * pav.z);
                        const T jy = w * (pv.y - npy) - w * (D0.y * pav.x + D1.y * pav.y + D2.y * pav.z);
                        const T jz = w * (pv.z - npz) - w * (D0.z * pav.x + D1.z * pav.y + D2.z * pav.z);
                        s0 += wj * jx; s1 += wj * jy; s2 += wj * jz;
                        s3 -= wj * (D0.x * jx + D0.y * jy + D0.z * jz); s4 -= wj * (D1.x * jx + D1.y * jy + D1.z * jz); s5 -= wj * (D2.x * jx + D2.y * jy + D2.z * jz);
                        if (wj != T(0)) accDen += (double)(jx * jx + jy * jy + jz * jz);      // sum_u p_u (J^T J p)_u of this edge = |J p|^2 (o.t:2117-2122)
                    }
                    {   // its reverse (u -> v): only its J p reaches this vertex's Offset row; E_k from the slot (arap_cols of the neighbour's coefficients, formed once per step)
                        const T jx = w * (npx - pv.x) - w * (sl[b].e0x * nax + sl[b].e1x * nay + sl[b].e2x * naz);
                        const T jy = w * (npy - pv.y) - w * (sl[b].e0y * nax + sl[b].e1y * nay + sl[b].e2y * naz);
                        const T jz = w * (npz - pv.z) - w * (sl[b].e0z * nax + sl[b].e1z * nay + T(0) * naz);
                        s0 -= wj * jx; s1 -= wj * jy; s2 -= wj * jz;
                    }
                }
            }
            if (ok[j]) {
                // per-vertex ("centred") part: the fitting term -- what arap_vertices<3> computes
                const V3<T> q{wf2[j] * pv.x, wf2[j] * pv.y, wf2[j] * pv.z};
                accDen += (double)(dot3(pv, q) + T(0));
                const V3<T> oO{q.x + s0, q.y + s1, q.z + s2}, oA{T(0) + s3, T(0) + s4, T(0) + s5};
                ap[j][0] = oO.x; ap[j][1] = oO.y; ap[j][2] = oO.z; ap[j][3] = oA.x; ap[j][4] = oA.y; ap[j][5] = oA.z;
                const V3<T> dgO = ld3(K.diag, i), dgA = ld3(K.diag + offA, i);
                const T m[6] = {aoGi(dgO.x), aoGi(dgO.y), aoGi(dgO.z), aoGi(dgA.x), aoGi(dgA.y), aoGi(dgA.z)};
#pragma unroll
                for (int c = 0; c < 6; ++c) {      // the expansion sums of arap_applySym: exact double products of M, r, A p
                    accNum += arap_dprod3(m[c], r[j][c], r[j][c]);
                    acc2 += arap_dprod3(m[c], r[j][c], ap[j][c]);
                    acc3 += arap_dprod3(m[c], ap[j][c], ap[j][c]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) ap[j][c] = 0;
            }
        }
        AO_MARK(0);      // gather
        // ---- the grid-wide sums: every workgroup posts, every workgroup reads all and adds in workgroup order -------------------------------------------------
        {
            double v4[4] = {accNum, accDen, acc2, acc3};
#pragma unroll
            for (int q = 0; q < 4; ++q) { v4[q] = ocWaveSum63(v4[q]); if (lane == kWave - 1) red[q * kAoWaves + wave] = v4[q]; }
            __syncthreads();
            oc_u64* const slotPar = K.S.slots + (long)(tag & 1u) * K.G * 8;
            if (tid < 8) {
                double s = 0;
                for (int wv = 0; wv < kAoWaves; ++wv) s += red[(tid >> 1) * kAoWaves + wv];
                const oc_u64 b = (oc_u64)__double_as_longlong(s);
                ocStore(slotPar + (long)g * 8 + tid, tag, (tid & 1) ? (unsigned)(b >> 32) : (unsigned)b);
            }
            constexpr int kPer = kAoMaxGrid * 8 / kAoBlock;      // a lane's (up to 4) requests are in flight together
            const int nW = K.G * 8;
            oc_u64 wv[kPer];
            auto fetchSums = [&]() {
                bool ready = true;
#pragma unroll
                for (int u = 0; u < kPer; ++u) { const int i = tid + u * kAoBlock; wv[u] = ocLoad(slotPar + (i < nW ? i : tid)); }
#pragma unroll
                for (int u = 0; u < kPer; ++u) ready = ready && (unsigned)(wv[u] >> 32) == tag;
                return ready;
            };
            if (tid < nW && !fetchSums()) {
                const long long t0 = wall_clock64();
                unsigned spins = 0;
                for (;;) {
                    __builtin_amdgcn_s_sleep(1);
                    if (fetchSums()) break;
                    if ((++spins & 31u) == 0) {
                        if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        if (wall_clock64() - t0 > to) { __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < kPer; ++u) { const int i = tid + u * kAoBlock; if (i < nW) W1[i] = (unsigned)wv[u]; }
            __syncthreads();
            if (tid < 4) { double s = 0; for (int m = 0; m < K.G; ++m) s += ocJoin(W1[m * 8 + 2 * tid], W1[m * 8 + 2 * tid + 1]); GS[tid] = s; }
            if (tid == 0) badL = __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
        AO_MARK(1);      // sums (incl. waiting for the slowest workgroup's gather)
        const double aNumD = GS[0], aDenD = GS[1], sum2 = GS[2], sum3 = GS[3];
        if (badL) { failed = true; break; }
        if (K.trace && g == 0 && tid == 0) { K.trace[4 * k] = aNumD; K.trace[4 * k + 1] = aDenD; K.trace[4 * k + 2] = sum2; K.trace[4 * k + 3] = sum3; }
        const T aNum = (T)aNumD, aDen = (T)aDenD;
        const T alpha = (aDen > T(0)) ? aNum / aDen : T(0);                                            // solver.t:456-459
        const double bNumD = fmax(aNumD - 2.0 * (double)alpha * sum2 + (double)alpha * (double)alpha * sum3, 0.0);
        const T beta = (aNum > T(0)) ? (T)bNumD / aNum : T(0);                                         // solver.t:544-547
        const bool last = k + 1 == K.L;
        // ---- PCGStep2 + PCGStep3 (arap_flatStepRec's arithmetic): delta += alpha p; r -= alpha A p; p = M r + beta p ------------------------------------------------
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            if (!ok[j]) continue;
            const long i = vid[j];
            V3<T> dO = ld3(K.delta, i), dA = ld3(K.delta + offA, i);
            dO.x = dO.x + alpha * p[j][0]; dO.y = dO.y + alpha * p[j][1]; dO.z = dO.z + alpha * p[j][2];
            dA.x = dA.x + alpha * p[j][3]; dA.y = dA.y + alpha * p[j][4]; dA.z = dA.z + alpha * p[j][5];
            K.delta[3 * i] = dO.x; K.delta[3 * i + 1] = dO.y; K.delta[3 * i + 2] = dO.z;
            K.delta[offA + 3 * i] = dA.x; K.delta[offA + 3 * i + 1] = dA.y; K.delta[offA + 3 * i + 2] = dA.z;
            if (last) {      // PCGLinearUpdate: X += delta (nothing else survives the last iteration)
                K.XO[3 * i] += dO.x; K.XO[3 * i + 1] += dO.y; K.XO[3 * i + 2] += dO.z;
                K.XA[3 * i] += dA.x; K.XA[3 * i + 1] += dA.y; K.XA[3 * i + 2] += dA.z;
                continue;
            }
            const V3<T> dgO = ld3(K.diag, i), dgA = ld3(K.diag + offA, i);
            const T m[6] = {aoGi(dgO.x), aoGi(dgO.y), aoGi(dgO.z), aoGi(dgA.x), aoGi(dgA.y), aoGi(dgA.z)};
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const T rn = r[j][c] - alpha * ap[j][c];
                const T z = m[c] * rn;
                r[j][c] = rn;
                p[j][c] = z + beta * p[j][c];
            }
        }
        AO_MARK(2);      // update
        if (!last) {      // (every wave of the workgroup is past its gather: the sums' barriers lie in between)
            shareAndPublish(tag + 1u);
            AO_MARK(3);      // share + publish
            fetchHalo(tag + 1u);
            AO_MARK(4);      // halo fetch
            __syncthreads();
            AO_MARK(5);      // barrier
        }
    }
    (void)failed;
    if (K.prof && tid == 0) for (int q = 0; q < 6; ++q) K.prof[(long)g * 8 + q] = tPh[q];
}

// Behind the persistent launch: tell the host if a wait timed out (then nothing was applied: every workgroup left before its last iteration).
__global__ void arap_relayBad(const int* __restrict__ bad, int* hostErr) {
    if (threadIdx.x == 0 && __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) __hip_atomic_store(hostErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace
}  // namespace optamd
