// Functor-driven kernel set for energies on meshes: residuals per vertex plus residuals per hyperedge of a graph.
//
// Counterpart of stencil_engine.h for the graph domain (reference o.t:2092-2126 applyJTJ_Graph, :2228-2253 evalJTF_Graph;
// solverGPUGaussNewton.t:687-706).  The energy is a device functor that writes its vertex residuals and its edge residuals once
// against a scalar type S; the engine instantiates them with plain and dual numbers:
//   vertex pass (one thread per vertex): OVERWRITES the vertex's rows with the contribution of its own residuals (+ CtC p),
//   edge pass (one thread per hyperedge): S = Dual<T, V*K (+1)> over the V vertices of the edge; J^T (J p) is scattered with one
//                atomic per (vertex, unknown) -- wave-aggregated for the first vertex, whose edges arrive grouped (OptGraph.h:64-76).
// Like the reference's graph kernels the scatter order is not fixed, so sums can differ in the last bits from run to run.
// The functor G provides (constexpr / static): NIMG, K, imgOf(k), chOf(k), channels(img), V, RV, RE, edgeDepends(ri, j);
// members N, nE, vidx[V], X[NIMG]; host bindParams(void**), unknownParam(img).
#pragma once
#include "stencil_engine.h"
#include "graph_common.h"

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

// MODE 0: cost, 1: model cost (vec = delta), 2: J^T F + diag, 3: J^T J vec
template <class T, class G, int MODE>
__global__ __launch_bounds__(kBlock) void ge_vertices(G g, const T* __restrict__ vec, GOff<G> vo, T* __restrict__ out, T* __restrict__ diag, const T* __restrict__ CtC,
                                                       double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    double acc = 0;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < g.N; v += (long)gridDim.x * blockDim.x) {
        typedef VertexCtx<T, G, MODE == 1 || MODE == 3, MODE >= 2> Ctx;
        typedef typename Ctx::S S;
        S r[G::RV > 0 ? G::RV : 1];
        g.template vertexResiduals<S>(Ctx{g, v, vec, vo}, v, r);
        if constexpr (MODE == 0) { T s = 0; for (int i = 0; i < G::RV; ++i) s += r[i] * r[i]; acc += (double)(T(0.5) * s); }
        else if constexpr (MODE == 1) { T s = 0; for (int i = 0; i < G::RV; ++i) { const T m = r[i].v + r[i].d[0]; s += m * m; } acc += (double)(T(0.5) * s); }
        else {
#pragma unroll
            for (int k = 0; k < G::K; ++k) {
                const long u = unknownIndex<T, G>(vo, v, k);
                T gk = 0, dk = 0;
#pragma unroll
                for (int i = 0; i < G::RV; ++i) {
                    if (MODE == 2) { gk += r[i].d[k] * r[i].v; dk += r[i].d[k] * r[i].d[k]; }
                    else gk += r[i].d[1 + k] * r[i].d[0];
                }
                if (MODE == 2) { out[u] = -gk; diag[u] = dk; }
                else { if (CtC) gk += CtC[u] * vec[u]; out[u] = gk; acc += (double)(vec[u] * gk); }
            }
        }
    }
    if (MODE != 2) { const double t = blockReduceSum(acc, scratch); if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t; }
}

template <class T, class G, int MODE>
__global__ __launch_bounds__(kBlock) void ge_edges(G g, const T* __restrict__ vec, GOff<G> vo, T* __restrict__ out, T* __restrict__ diag, double* __restrict__ partials) {
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
    }
    int vgrid() const { return (int)std::max<long>(1, std::min<long>((g.N + kBlock - 1) / kBlock, kMaxPartials / 2)); }
    void bind(void** p, LaunchCtx&) override { g.bindParams(p); }
    T* unknownPtr(int img) const override { return const_cast<T*>(g.X[img]); }
    void evalCost(Reduction& out, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(g.nE, cus);
        { ScopedKernel k(ctx, "computeCost"); ge_vertices<T, G, 0><<<gv, kBlock, 0, ctx.stream>>>(g, nullptr, vo, nullptr, nullptr, nullptr, out.partials); }
        { ScopedKernel k(ctx, "computeCost_Graph"); ge_edges<T, G, 0><<<ge, kBlock, 0, ctx.stream>>>(g, nullptr, vo, nullptr, nullptr, out.partials + gv); }
        out.n = gv + ge;
    }
    void evalJTF(T* r, T* diag, LaunchCtx& ctx) override {
        { ScopedKernel k(ctx, "PCGInit1"); ge_vertices<T, G, 2><<<vgrid(), kBlock, 0, ctx.stream>>>(g, nullptr, vo, r, diag, nullptr, nullptr); }
        { ScopedKernel k(ctx, "PCGInit1_Graph"); ge_edges<T, G, 2><<<edgeGrid(g.nE, cus), kBlock, 0, ctx.stream>>>(g, nullptr, vo, r, diag, nullptr); }
    }
    void applyJTJ(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(g.nE, cus);
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
