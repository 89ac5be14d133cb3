// Functor-driven kernel set for energies on meshes: residuals per vertex plus residuals per hyperedge of a graph.
//
// Counterpart of stencil_engine.h for the graph domain (reference o.t:2092-2126 applyJTJ_Graph, :2228-2253 evalJTF_Graph;
// solverGPUGaussNewton.t:687-706).  The energy is a device functor that writes its vertex residuals and its edge residuals once
// against a scalar type S; the engine instantiates them with plain and dual numbers:
//   edge pass (one thread per hyperedge): S = Dual<T, V*K (+1)> over the V vertices of the edge; its J^T (J p) contributions go into one
//                K-scalar record per (hyperedge, slot),
//   vertex pass (one thread per vertex): the contribution of the vertex's own residuals (+ CtC p) plus the records of every (hyperedge, slot)
//                the vertex occupies, in ascending (slot, hyperedge) order from incidence lists built once per graph by a counting sort.
// No atomics: unlike the reference's graph kernels (one atomic per (vertex, unknown), unordered) a solve is bit-reproducible.
// OPT_AMD_GRAPH_GATHER=0 selects the scatter formulation instead (wave-aggregated atomics for the first vertex, whose edges arrive
// grouped, OptGraph.h:64-76; plain atomics for the others).
// The functor G provides (constexpr / static): NIMG, K, imgOf(k), chOf(k), channels(img), V, RV, RE, edgeDepends(ri, j);
// members N, nE, vidx[V], X[NIMG]; host bindParams(void**), unknownParam(img).
#pragma once
#include "stencil_engine.h"
#include "graph_common.h"
#include <hipcub/hipcub.hpp>

namespace optamd {

template <class G> struct GOff { long o[G::NIMG]; };

template <class T, class G> __device__ __forceinline__ long unknownIndex(const GOff<G>& vo, long vertex, int k) {
    return vo.o[G::imgOf(k)] + vertex * G::channels(G::imgOf(k)) + G::chOf(k);
}
// unknown k of one vertex.  DIR: slot 0 = component of the solver vector `vec`; SEED: slots DIR.. = d/d(unknown k)
template <class T, class G, bool DIR, bool SEED>
struct VertexCtx {
    static constexpr int N = (DIR ? 1 : 0) + (SEED ? G::K : 0);
    typedef typename std::conditional<N == 0, T, Dual<T, (N > 0 ? N : 1)>>::type S;
    const G& g; long v; const T* vec; const GOff<G>& vo;
    __device__ __forceinline__ S operator()(int k) const {
        const T x = g.X[G::imgOf(k)][v * G::channels(G::imgOf(k)) + G::chOf(k)];
        if constexpr (N == 0) return x;
        else { S r(x); if (DIR) r.d[0] = vec[unknownIndex<T, G>(vo, v, k)]; if (SEED) r.d[(DIR ? 1 : 0) + k] = T(1); return r; }
    }
};
// unknown k of vertex j of one hyperedge
template <class T, class G, bool DIR, bool SEED>
struct EdgeCtx {
    static constexpr int N = (DIR ? 1 : 0) + (SEED ? G::V * G::K : 0);
    typedef typename std::conditional<N == 0, T, Dual<T, (N > 0 ? N : 1)>>::type S;
    const G& g; long vid[G::V]; const T* vec; const GOff<G>& vo;
    __device__ __forceinline__ S operator()(int j, int k) const {
        const T x = g.X[G::imgOf(k)][vid[j] * G::channels(G::imgOf(k)) + G::chOf(k)];
        if constexpr (N == 0) return x;
        else { S r(x); if (DIR) r.d[0] = vec[unknownIndex<T, G>(vo, vid[j], k)]; if (SEED) r.d[(DIR ? 1 : 0) + j * G::K + k] = T(1); return r; }
    }
};

// ---- incidence lists: for every vertex the ids j * nE + e of the (hyperedge e, slot j) pairs it occupies, ascending ----------------------
template <int V> struct SlotIdx { const int* p[V]; };
struct Incidence { const int* off; const int* idx; };       // off == nullptr: scatter mode
template <int V>
__global__ __launch_bounds__(kBlock) void inc_count(SlotIdx<V> vi, int nE, int* __restrict__ deg, unsigned long long* __restrict__ checksum) {
    unsigned long long acc = 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nE; e += gridDim.x * blockDim.x)
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int v = vi.p[j][e];
            if (deg) atomicAdd(deg + v, 1);
            acc += ((unsigned long long)(unsigned)v * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)(j * (long)nE + e) * 0x165667B19E3779F9ull);
        }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
    if (checksum && (threadIdx.x & (kWave - 1)) == 0) atomicAdd(checksum, acc);     // integer sum: order-independent
}
template <int V>
__global__ __launch_bounds__(kBlock) void inc_fill(SlotIdx<V> vi, int nE, const int* __restrict__ off, int* __restrict__ cur, int* __restrict__ idx) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nE; e += gridDim.x * blockDim.x)
#pragma unroll
        for (int j = 0; j < V; ++j) { const int v = vi.p[j][e]; idx[off[v] + atomicAdd(cur + v, 1)] = j * nE + e; }
}
template <int V>
__global__ __launch_bounds__(kBlock) void inc_sort(long N, const int* __restrict__ off, int* __restrict__ idx) {   // per-vertex insertion sort: fixed summation order
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < N; v += (long)gridDim.x * blockDim.x) {
        const int b = off[v], e = off[v + 1];
        for (int i = b + 1; i < e; ++i) { const int x = idx[i]; int j = i - 1; while (j >= b && idx[j] > x) { idx[j + 1] = idx[j]; --j; } idx[j + 1] = x; }
    }
}

// MODE 0: cost, 1: model cost (vec = delta), 2: J^T F + diag, 3: J^T J vec.  LANES > 1 (gather mode): LANES adjacent lanes share one vertex;
// lane 0 evaluates the vertex's own residuals, every lane walks every LANES-th record of the incidence list, and the partial sums are
// folded with shuffles in a fixed order (more independent loads in flight than one thread walking ~24 records).
#ifndef GE_GATHER_LANES
#define GE_GATHER_LANES 4
#endif
template <class T, class G, int MODE, int LANES = 1>
__global__ __launch_bounds__(kBlock) void ge_vertices(G g, const T* __restrict__ vec, GOff<G> vo, T* __restrict__ out, T* __restrict__ diag, const T* __restrict__ CtC,
                                                       double* __restrict__ partials, Incidence inc = Incidence{nullptr, nullptr}, const T* __restrict__ rec = nullptr) {
    __shared__ double scratch[kBlock / kWave + 1];
    double acc = 0;
    const int sub = LANES > 1 ? threadIdx.x % LANES : 0;
    for (long v = (blockIdx.x * (long)blockDim.x + threadIdx.x) / LANES; v < g.N; v += (long)gridDim.x * blockDim.x / LANES) {
        typedef VertexCtx<T, G, MODE == 1 || MODE == 3, MODE >= 2> Ctx;
        typedef typename Ctx::S S;
        S r[G::RV > 0 ? G::RV : 1];
        if (sub == 0) g.template vertexResiduals<S>(Ctx{g, v, vec, vo}, v, r);
        if constexpr (MODE == 0) { T s = 0; for (int i = 0; i < G::RV; ++i) s += r[i] * r[i]; acc += (double)(T(0.5) * s); }
        else if constexpr (MODE == 1) { T s = 0; for (int i = 0; i < G::RV; ++i) { const T m = r[i].v + r[i].d[0]; s += m * m; } acc += (double)(T(0.5) * s); }
        else {
            T gs[G::K], ds[G::K];
#pragma unroll
            for (int k = 0; k < G::K; ++k) {
                T gk = 0, dk = 0;
                if (sub == 0) {
#pragma unroll
                    for (int i = 0; i < G::RV; ++i) {
                        if (MODE == 2) { gk += r[i].d[k] * r[i].v; dk += r[i].d[k] * r[i].d[k]; }
                        else gk += r[i].d[1 + k] * r[i].d[0];
                    }
                    if (MODE == 3) { const long u = unknownIndex<T, G>(vo, v, k); if (CtC) gk += CtC[u] * vec[u]; acc += (double)(vec[u] * gk); }   // the edges' share of p . A p is |J p|^2, summed by the edge pass
                }
                gs[k] = gk; ds[k] = dk;
            }
            if (inc.off) {      // gather mode: add the records of the (hyperedge, slot) pairs of this vertex, ascending
                constexpr int KK = MODE == 2 ? 2 * G::K : G::K;
                for (int t = inc.off[v] + sub, te = inc.off[v + 1]; t < te; t += LANES) {
                    const T* q = rec + (long)inc.idx[t] * KK;
#pragma unroll
                    for (int k = 0; k < G::K; ++k) { gs[k] += q[k]; if (MODE == 2) ds[k] += q[G::K + k]; }
                }
                if (LANES > 1) {
#pragma unroll
                    for (int off = LANES / 2; off > 0; off >>= 1)
#pragma unroll
                        for (int k = 0; k < G::K; ++k) { gs[k] += __shfl_down(gs[k], off, LANES); if (MODE == 2) ds[k] += __shfl_down(ds[k], off, LANES); }
                }
            }
            if (sub == 0) {
#pragma unroll
                for (int k = 0; k < G::K; ++k) {
                    const long u = unknownIndex<T, G>(vo, v, k);
                    if (MODE == 2) { out[u] = -gs[k]; diag[u] = ds[k]; } else out[u] = gs[k];
                }
            }
        }
    }
    if (MODE != 2) { const double t = blockReduceSum(acc, scratch); if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t; }
}

template <class T, class G, int MODE>
__global__ __launch_bounds__(kBlock) void ge_edges(G g, const T* __restrict__ vec, GOff<G> vo, T* __restrict__ out, T* __restrict__ diag, double* __restrict__ partials,
                                                    T* __restrict__ rec = nullptr) {
    __shared__ double scratch[kBlock / kWave + 1];
    double acc = 0;
    const long nE = g.nE, nLoop = ((nE + kBlock - 1) / kBlock) * kBlock;       // whole waves stay in the loop: the aggregation shuffles need them
    for (long e0 = blockIdx.x * (long)blockDim.x + threadIdx.x; e0 < nLoop; e0 += (long)gridDim.x * blockDim.x) {
        const bool ok = e0 < nE;
        const long e = ok ? e0 : 0;
        typedef EdgeCtx<T, G, MODE == 1 || MODE == 3, MODE >= 2> Ctx;
        typedef typename Ctx::S S;
        Ctx X{g, {}, vec, vo};
#pragma unroll
        for (int j = 0; j < G::V; ++j) X.vid[j] = g.vidx[j][e];
        S r[G::RE];
        g.template edgeResiduals<S>(X, e, r);
        if constexpr (MODE == 0) { T s = 0; for (int i = 0; i < G::RE; ++i) s += r[i] * r[i]; if (ok) acc += (double)(T(0.5) * s); }
        else if constexpr (MODE == 1) { T s = 0; for (int i = 0; i < G::RE; ++i) { const T m = r[i].v + r[i].d[0]; s += m * m; } if (ok) acc += (double)(T(0.5) * s); }
        else {
            if (MODE == 3 && ok) { T s = 0; for (int i = 0; i < G::RE; ++i) s += r[i].d[0] * r[i].d[0]; acc += (double)s; }   // sum_u p_u (J^T J p)_u of this edge = |J p|^2 (o.t:2117-2122)
#pragma unroll
            for (int j = 0; j < G::V; ++j) {
#pragma unroll
                for (int k = 0; k < G::K; ++k) {
                    T gk = 0, dk = 0; bool any = false;
#pragma unroll
                    for (int i = 0; i < G::RE; ++i) {
                        if (!G::edgeDepends(i, j, k)) continue;
                        any = true;
                        const T d = r[i].d[(MODE == 3 ? 1 : 0) + j * G::K + k];
                        if (MODE == 2) { gk += d * r[i].v; dk += d * d; } else gk += d * r[i].d[0];
                    }
                    if (rec) {      // gather mode: record (slot j, hyperedge e), K (J^T J p) or 2 K (J^T F, then diag) scalars
                        if (ok) { constexpr int KK = MODE == 2 ? 2 * G::K : G::K; T* q = rec + ((long)j * nE + e) * KK; q[k] = gk; if (MODE == 2) q[G::K + k] = dk; }
                        continue;
                    }
                    if (!any) continue;
                    const long u = unknownIndex<T, G>(vo, X.vid[j], k);
                    if (j == 0) { segmentedAtomicAdd(out, u, MODE == 2 ? -gk : gk, ok); if (MODE == 2) segmentedAtomicAdd(diag, u, dk, ok); }
                    else if (ok) { plainAtomicAdd(out + u, MODE == 2 ? -gk : gk); if (MODE == 2) plainAtomicAdd(diag + u, dk); }
                }
            }
        }
    }
    if (MODE != 2) { const double t = blockReduceSum(acc, scratch); if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t; }
}

template <class T, class G>
struct GraphOps : EnergyOps<T> {
    G g{};
    GOff<G> vo{};
    int cus = 256;
    GraphOps(const unsigned* dims, bool usePre) {
        g.N = dims[0];
        this->usePreconditioner = usePre; this->usesGraph = true;
        for (int i = 0; i < G::NIMG; ++i) { vo.o[i] = this->nScalars; this->addUnknown(G::unknownParam(i), g.N, G::channels(i)); }
        int dev = 0; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (const char* e = getenv("OPT_AMD_GRAPH_GATHER")) useGather = atoi(e) != 0;
    }
    int vgrid(int lanes = 1) const { return (int)std::max<long>(1, std::min<long>((g.N * lanes + kBlock - 1) / kBlock, kMaxPartials / 2)); }
    // gather mode state: incidence lists (rebuilt when the graph arrays change: pointers, count or an order-independent checksum) and the record buffer
    bool useGather = true;
    int *incOff = nullptr, *incIdx = nullptr, *cursors = nullptr; T* rec = nullptr; long recCapacity = 0;
    void* scanTemp = nullptr; size_t scanTempBytes = 0; unsigned long long* dChecksum = nullptr;
    const int* incV[G::V] = {}; int incNE = -1; unsigned long long incSum = 0; bool incValid = false;
    ~GraphOps() override { for (void* q : {(void*)incOff, (void*)incIdx, (void*)cursors, (void*)rec, scanTemp, (void*)dChecksum}) if (q) (void)hipFree(q); }
    Incidence incidence() const { return useGather ? Incidence{incOff, incIdx} : Incidence{nullptr, nullptr}; }
    void ensureIncidence(LaunchCtx& ctx) {
        hipStream_t st = ctx.stream;
        SlotIdx<G::V> vi; bool same = incValid && incNE == g.nE;
        for (int j = 0; j < G::V; ++j) { vi.p[j] = g.vidx[j]; same = same && incV[j] == g.vidx[j]; }
        if (!dChecksum) HIP_CHECK(hipMalloc((void**)&dChecksum, 8));
        HIP_CHECK(hipMemsetAsync(dChecksum, 0, 8, st));
        const int ge = edgeGrid(g.nE, cus);
        inc_count<G::V><<<ge, kBlock, 0, st>>>(vi, g.nE, nullptr, dChecksum);
        unsigned long long sum = 0;
        HIP_CHECK(hipMemcpyAsync(&sum, dChecksum, 8, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipStreamSynchronize(st));
        if (same && sum == incSum) return;
        ScopedKernel k(ctx, "buildIncidenceLists");
        for (void* q : {(void*)incOff, (void*)incIdx, (void*)cursors}) if (q) HIP_CHECK(hipFree(q));
        const size_t nv = (size_t)g.N + 1, nInc = (size_t)std::max(1, g.nE) * G::V;
        HIP_CHECK(hipMalloc((void**)&incOff, nv * 4)); HIP_CHECK(hipMalloc((void**)&cursors, nv * 4)); HIP_CHECK(hipMalloc((void**)&incIdx, nInc * 4));
        if ((long)nInc > recCapacity) { if (rec) HIP_CHECK(hipFree(rec)); HIP_CHECK(hipMalloc((void**)&rec, nInc * 2 * G::K * sizeof(T))); recCapacity = (long)nInc; }
        HIP_CHECK(hipMemsetAsync(cursors, 0, nv * 4, st));
        inc_count<G::V><<<ge, kBlock, 0, st>>>(vi, g.nE, cursors, nullptr);
        size_t need = 0;
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, cursors, incOff, (int)nv, st));
        if (need > scanTempBytes) { if (scanTemp) HIP_CHECK(hipFree(scanTemp)); HIP_CHECK(hipMalloc(&scanTemp, need)); scanTempBytes = need; }
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTemp, need, cursors, incOff, (int)nv, st));
        HIP_CHECK(hipMemsetAsync(cursors, 0, nv * 4, st));
        inc_fill<G::V><<<ge, kBlock, 0, st>>>(vi, g.nE, incOff, cursors, incIdx);
        inc_sort<G::V><<<vgrid(), kBlock, 0, st>>>(g.N, incOff, incIdx);
        for (int j = 0; j < G::V; ++j) incV[j] = g.vidx[j];
        incNE = g.nE; incSum = sum; incValid = true;
    }
    void bind(void** p, LaunchCtx& ctx) override {
        g.bindParams(p);
        if (useGather && (long)g.nE * G::V > 0x7fffffffL) useGather = false;      // incidence ids j * nE + e are 32-bit: beyond that, scatter
        if (useGather) ensureIncidence(ctx);
    }
    T* unknownPtr(int img) const override { return const_cast<T*>(g.X[img]); }
    void evalCost(Reduction& out, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(g.nE, cus);
        { ScopedKernel k(ctx, "computeCost"); ge_vertices<T, G, 0><<<gv, kBlock, 0, ctx.stream>>>(g, nullptr, vo, nullptr, nullptr, nullptr, out.partials); }
        { ScopedKernel k(ctx, "computeCost_Graph"); ge_edges<T, G, 0><<<ge, kBlock, 0, ctx.stream>>>(g, nullptr, vo, nullptr, nullptr, out.partials + gv); }
        out.n = gv + ge;
    }
    void evalJTF(T* r, T* diag, LaunchCtx& ctx) override {
        if (useGather) {    // records first, then the vertex pass gathers them
            { ScopedKernel k(ctx, "PCGInit1_Graph"); ge_edges<T, G, 2><<<edgeGrid(g.nE, cus), kBlock, 0, ctx.stream>>>(g, nullptr, vo, r, diag, nullptr, rec); }
            { ScopedKernel k(ctx, "PCGInit1"); ge_vertices<T, G, 2, GE_GATHER_LANES><<<vgrid(GE_GATHER_LANES), kBlock, 0, ctx.stream>>>(g, nullptr, vo, r, diag, nullptr, nullptr, incidence(), rec); }
            return;
        }
        { ScopedKernel k(ctx, "PCGInit1"); ge_vertices<T, G, 2><<<vgrid(), kBlock, 0, ctx.stream>>>(g, nullptr, vo, r, diag, nullptr, nullptr); }
        { ScopedKernel k(ctx, "PCGInit1_Graph"); ge_edges<T, G, 2><<<edgeGrid(g.nE, cus), kBlock, 0, ctx.stream>>>(g, nullptr, vo, r, diag, nullptr); }
    }
    void applyJTJ(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx) override {
        const int gv = vgrid(useGather ? GE_GATHER_LANES : 1), ge = edgeGrid(g.nE, cus);
        if (useGather) {
            { ScopedKernel k(ctx, "PCGStep1_Graph"); ge_edges<T, G, 3><<<ge, kBlock, 0, ctx.stream>>>(g, v, vo, out, nullptr, dot ? dot->partials + gv : nullptr, rec); }
            { ScopedKernel k(ctx, "PCGStep1"); ge_vertices<T, G, 3, GE_GATHER_LANES><<<gv, kBlock, 0, ctx.stream>>>(g, v, vo, out, nullptr, CtC, dot ? dot->partials : nullptr, incidence(), rec); }
            if (dot) dot->n = gv + ge;
            return;
        }
        { ScopedKernel k(ctx, "PCGStep1"); ge_vertices<T, G, 3><<<gv, kBlock, 0, ctx.stream>>>(g, v, vo, out, nullptr, CtC, dot ? dot->partials : nullptr); }
        { ScopedKernel k(ctx, "PCGStep1_Graph"); ge_edges<T, G, 3><<<ge, kBlock, 0, ctx.stream>>>(g, v, vo, out, nullptr, dot ? dot->partials + gv : nullptr); }
        if (dot) dot->n = gv + ge;
    }
    void evalModelCost(const T* delta, Reduction& out, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(g.nE, cus);
        { ScopedKernel k(ctx, "computeModelCost"); ge_vertices<T, G, 1><<<gv, kBlock, 0, ctx.stream>>>(g, delta, vo, nullptr, nullptr, nullptr, out.partials); }
        { ScopedKernel k(ctx, "computeModelCost_Graph"); ge_edges<T, G, 1><<<ge, kBlock, 0, ctx.stream>>>(g, delta, vo, nullptr, nullptr, out.partials + gv); }
        out.n = gv + ge;
    }
};

}  // namespace optamd
