// Graph-domain energies: arap_mesh_deformation (BASELINE config 4) and curveFitting (the reference's
// tests/minimal_graph_only known-answer test).
//
// Reference behaviour (API/src/o.t:2092-2126, 2228-2253; solverGPUGaussNewton.t:687-706): one thread per
// hyperedge evaluates the residual's Jacobian row block, forms J p, and scatters J^T (J p) into the unknown
// vector with one atomic per (vertex, channel); the per-vertex ("centred") kernel runs first and OVERWRITES,
// the edge kernel then accumulates on top (solver.t:1032-1036, 1062-1065).
//
// MI355X design: the scatter is done with wave-aggregated atomics.  Edges arrive grouped by head vertex
// (examples/shared/OptGraph.h:64-76), so inside a wave64 the lanes that target the same head form contiguous
// runs; a 6-step segmented shuffle reduction folds each run into its first lane, which issues ONE hardware
// f32/f64 atomic (gfx950 has both natively -- no CAS loop as in util.t:574-597).  Runs need not be sorted for
// correctness, only for the aggregation to pay off.  Tail-vertex targets are irregular and use plain atomics.
#include "energy.h"

namespace optamd {
namespace {

// val summed over each contiguous run of equal `key` inside the wave; then one atomic per run.
template <class T>
__device__ __forceinline__ void segmentedAtomicAdd(T* __restrict__ base, long key, T val, bool active) {
    const int lane = threadIdx.x & (kWave - 1);
    const long k = active ? key : -1 - lane;              // inactive lanes get unique keys: never merged
    const long prev = __shfl_up(k, 1, kWave);
    const bool head = (lane == 0) || (prev != k);
    const unsigned long long heads = __ballot(head);
    const unsigned long long above = (lane == kWave - 1) ? 0ull : (heads >> (lane + 1));
    const int runEnd = above ? lane + 1 + __builtin_ctzll(above) : kWave;   // first lane of the next run
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const T other = __shfl_down(val, off, kWave);
        if (lane + off < runEnd) val += other;             // only lanes of my own contiguous run are folded in
    }
    if (active && head) unsafeAtomicAdd(base + key, val);
}
template <class T> __device__ __forceinline__ void plainAtomicAdd(T* addr, T val) { unsafeAtomicAdd(addr, val); }

// ------------------------------------------------------------------------------------------------------------------
// curveFitting.t: unknown funcParams(a,b) over U, data(x,y) over N, edge (d,p): r = y - (a cos(b x) + b sin(a x))
template <class T>
struct CFArgs { long N, U; const T* funcParams; const T* data; int nE; const int* dIdx; const int* pIdx; };

template <class T>
__device__ __forceinline__ void cf_eval(const CFArgs<T>& A, int e, T& res, T& ja, T& jb, long& pp) {
    const long d = A.dIdx[e]; pp = A.pIdx[e];
    const T x = A.data[2 * d], y = A.data[2 * d + 1];
    const T a = A.funcParams[2 * pp], b = A.funcParams[2 * pp + 1];
    T sbx, cbx, sax, cax; sincosT(b * x, &sbx, &cbx); sincosT(a * x, &sax, &cax);
    res = y - (a * cbx + b * sax);
    ja = -(cbx + b * x * cax);           // d res / d a
    jb = -(-a * x * sbx + sax);          // d res / d b
}
// MODE 0: cost, 1: model cost (delta), 2: evalJTF scatter (r -= J^T F, diag += J^2), 3: applyJTJ scatter
template <class T, int MODE>
__global__ __launch_bounds__(kBlock) void cf_edges(CFArgs<T> A, const T* __restrict__ v, T* __restrict__ out, T* __restrict__ out2, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    double acc = 0;
    const int nIter = (A.nE + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
    for (int it = 0; it < nIter; ++it) {   // uniform trip count: the segmented reduction needs the whole wave
        const int e = (it * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        const bool ok = e < A.nE;
        T res = 0, ja = 0, jb = 0; long pp = 0;
        if (ok) cf_eval(A, e, res, ja, jb, pp);
        if (MODE == 0) { if (ok) acc += (double)(T(0.5) * res * res); }
        else if (MODE == 1) { if (ok) { const T m = res + ja * v[2 * pp] + jb * v[2 * pp + 1]; acc += (double)(T(0.5) * m * m); } }
        else if (MODE == 2) {
            segmentedAtomicAdd(out, 2 * pp, -(ja * res), ok); segmentedAtomicAdd(out, 2 * pp + 1, -(jb * res), ok);
            segmentedAtomicAdd(out2, 2 * pp, ja * ja, ok); segmentedAtomicAdd(out2, 2 * pp + 1, jb * jb, ok);
        } else {
            const T jp = ok ? ja * v[2 * pp] + jb * v[2 * pp + 1] : T(0);
            segmentedAtomicAdd(out, 2 * pp, ja * jp, ok); segmentedAtomicAdd(out, 2 * pp + 1, jb * jp, ok);
            if (ok) acc += (double)(jp * jp);
        }
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
}
// centred stand-in (o.t:1972-1982): zero residual on the unknown index space -> overwrite with zeros / CtC*p
template <class T>
__global__ __launch_bounds__(kBlock) void zeroOrCtC(T* __restrict__ out, T* __restrict__ out2, const T* __restrict__ v, const T* __restrict__ CtC, long n, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    double acc = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        T o = 0;
        if (CtC) { o = CtC[i] * v[i]; acc += (double)(v[i] * o); }
        out[i] = o; if (out2) out2[i] = 0;
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
}

inline int edgeGrid(long nE, int cus) { return (int)std::max<long>(1, std::min<long>((nE + kBlock - 1) / kBlock, std::min<long>(kMaxPartials / 2, (long)cus * 8))); }

template <class T>
struct CurveFittingOps : EnergyOps<T> {
    CFArgs<T> A{};
    int cus = 256;
    CurveFittingOps(const unsigned* dims) {
        A.N = dims[0]; A.U = dims[1];
        this->usePreconditioner = true; this->usesGraph = true;
        this->addUnknown(0, A.U, 2);
        int dev = 0; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    void bind(void** p, LaunchCtx&) override {
        A.funcParams = (const T*)p[0]; A.data = (const T*)p[1];
        A.nE = *(const int*)p[2]; A.dIdx = (const int*)p[3]; A.pIdx = (const int*)p[4];   // Graph("G", 2, "d", {N}, 3, "p", {U}, 4); count read on the host (util.t:677-682)
    }
    T* unknownPtr(int) const override { return const_cast<T*>(A.funcParams); }
    int vgrid() const { return (int)std::max<long>(1, std::min<long>((this->nScalars + kBlock - 1) / kBlock, 64)); }
    void evalCost(Reduction& out, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "computeCost_Graph"); const int g = edgeGrid(A.nE, cus);
        cf_edges<T, 0><<<g, kBlock, 0, ctx.stream>>>(A, nullptr, nullptr, nullptr, out.partials); out.n = g;
    }
    void evalJTF(T* r, T* diag, LaunchCtx& ctx) override {
        { ScopedKernel k(ctx, "PCGInit1"); zeroOrCtC<T><<<vgrid(), kBlock, 0, ctx.stream>>>(r, diag, nullptr, nullptr, this->nScalars, nullptr); }
        { ScopedKernel k(ctx, "PCGInit1_Graph"); cf_edges<T, 2><<<edgeGrid(A.nE, cus), kBlock, 0, ctx.stream>>>(A, nullptr, r, diag, nullptr); }
    }
    void applyJTJ(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(A.nE, cus);
        { ScopedKernel k(ctx, "PCGStep1"); zeroOrCtC<T><<<gv, kBlock, 0, ctx.stream>>>(out, nullptr, v, CtC, this->nScalars, dot ? dot->partials : nullptr); }
        { ScopedKernel k(ctx, "PCGStep1_Graph"); cf_edges<T, 3><<<ge, kBlock, 0, ctx.stream>>>(A, v, out, nullptr, dot ? dot->partials + gv : nullptr); }
        if (dot) dot->n = gv + ge;
    }
    void evalModelCost(const T* delta, Reduction& out, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "computeModelCost_Graph"); const int g = edgeGrid(A.nE, cus);
        cf_edges<T, 1><<<g, kBlock, 0, ctx.stream>>>(A, delta, nullptr, nullptr, out.partials); out.n = g;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// arap_mesh_deformation.t:1-18.  Per vertex: r_fit = [C.x >= -999999.9] w_fit (O - C).  Per directed edge (v0,v1):
// r = w_reg [ (O_v0 - O_v1) - R3(a_v0)(U_v0 - U_v1) ];  Jacobian block [ w I, -w I, -w dR/da_k (U_v0 - U_v1) ].
template <class T>
struct ArapArgs {
    long N;
    const T* Offset; const T* Angle; const T* UrShape; const T* Constraints;
    T w_fit, w_reg; int nE; const int* v0; const int* v1;
};
template <class T> struct V3 { T x, y, z; };
template <class T> __device__ __forceinline__ V3<T> ld3(const T* p, long i) { return V3<T>{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
template <class T> __device__ __forceinline__ T dot3(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// rotated edge R3(a) u and the three derivative columns D_k = dR3/da_k u   (Rotate3D, reference lib.t:77-91)
template <class T>
__device__ __forceinline__ void arap_rot(const V3<T>& a, const V3<T>& u, V3<T>& Ru, V3<T>& D0, V3<T>& D1, V3<T>& D2) {
    T sa, ca, sb, cb, sg, cg;
    sincosT(a.x, &sa, &ca); sincosT(a.y, &sb, &cb); sincosT(a.z, &sg, &cg);
    Ru.x = (cg * cb) * u.x + (-sg * ca + cg * sb * sa) * u.y + (sg * sa + cg * sb * ca) * u.z;
    Ru.y = (sg * cb) * u.x + (cg * ca + sg * sb * sa) * u.y + (-cg * sa + sg * sb * ca) * u.z;
    Ru.z = (-sb) * u.x + (cb * sa) * u.y + (cb * ca) * u.z;
    // d/d alpha
    D0.x = (sg * sa + cg * sb * ca) * u.y + (sg * ca - cg * sb * sa) * u.z;
    D0.y = (-cg * sa + sg * sb * ca) * u.y + (-cg * ca - sg * sb * sa) * u.z;
    D0.z = (cb * ca) * u.y + (-cb * sa) * u.z;
    // d/d beta
    D1.x = (-cg * sb) * u.x + (cg * cb * sa) * u.y + (cg * cb * ca) * u.z;
    D1.y = (-sg * sb) * u.x + (sg * cb * sa) * u.y + (sg * cb * ca) * u.z;
    D1.z = (-cb) * u.x + (-sb * sa) * u.y + (-sb * ca) * u.z;
    // d/d gamma
    D2.x = (-sg * cb) * u.x + (-cg * ca - sg * sb * sa) * u.y + (cg * sa - sg * sb * ca) * u.z;
    D2.y = (cg * cb) * u.x + (-sg * ca + cg * sb * sa) * u.y + (sg * sa + cg * sb * ca) * u.z;
    D2.z = 0;
}

// per-vertex ("centred") kernels.  MODE 0 cost, 1 model cost, 2 evalJTF (overwrite r/diag), 3 applyJTJ (overwrite out)
template <class T, int MODE>
__global__ __launch_bounds__(kBlock) void arap_vertices(ArapArgs<T> A, const T* __restrict__ v, T* __restrict__ out, T* __restrict__ out2, const T* __restrict__ CtC,
                                                        double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    double acc = 0;
    const long offA = 3 * A.N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < A.N; i += (long)gridDim.x * blockDim.x) {
        const V3<T> c = ld3(A.Constraints, i), o = ld3(A.Offset, i);
        const bool valid = c.x >= T(-999999.9);                       // arap_mesh_deformation.t:13
        const T wf = valid ? A.w_fit : T(0);
        const V3<T> e{valid ? wf * (o.x - c.x) : T(0), valid ? wf * (o.y - c.y) : T(0), valid ? wf * (o.z - c.z) : T(0)};
        if (MODE == 0) acc += (double)(T(0.5) * dot3(e, e));
        else if (MODE == 1) { const V3<T> d = ld3(v, i); const V3<T> m{e.x + wf * d.x, e.y + wf * d.y, e.z + wf * d.z}; acc += (double)(T(0.5) * dot3(m, m)); }
        else if (MODE == 2) {
            out[3 * i] = -(wf * e.x); out[3 * i + 1] = -(wf * e.y); out[3 * i + 2] = -(wf * e.z);
            out2[3 * i] = wf * wf; out2[3 * i + 1] = wf * wf; out2[3 * i + 2] = wf * wf;
#pragma unroll
            for (int k = 0; k < 3; ++k) { out[offA + 3 * i + k] = 0; out2[offA + 3 * i + k] = 0; }
        } else {
            const V3<T> p = ld3(v, i); const V3<T> pa = ld3(v + offA, i);
            V3<T> q{wf * wf * p.x, wf * wf * p.y, wf * wf * p.z}, qa{0, 0, 0};
            if (CtC) {
                const V3<T> cO = ld3(CtC, i), cA = ld3(CtC + offA, i);
                q.x += cO.x * p.x; q.y += cO.y * p.y; q.z += cO.z * p.z; qa.x = cA.x * pa.x; qa.y = cA.y * pa.y; qa.z = cA.z * pa.z;
            }
            out[3 * i] = q.x; out[3 * i + 1] = q.y; out[3 * i + 2] = q.z;
            out[offA + 3 * i] = qa.x; out[offA + 3 * i + 1] = qa.y; out[offA + 3 * i + 2] = qa.z;
            acc += (double)(dot3(p, q) + dot3(pa, qa));
        }
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
}

template <class T, int MODE>
__global__ __launch_bounds__(kBlock) void arap_edges(ArapArgs<T> A, const T* __restrict__ v, T* __restrict__ out, T* __restrict__ out2, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    double acc = 0;
    const long offA = 3 * A.N;
    const int nIter = (A.nE + gridDim.x * blockDim.x - 1) / (gridDim.x * blockDim.x);
    for (int it = 0; it < nIter; ++it) {
        const int e = (it * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        const bool ok = e < A.nE;
        const long a0 = ok ? A.v0[e] : 0, a1 = ok ? A.v1[e] : 0;
        const V3<T> O0 = ld3(A.Offset, a0), O1 = ld3(A.Offset, a1), ang = ld3(A.Angle, a0), U0 = ld3(A.UrShape, a0), U1 = ld3(A.UrShape, a1);
        const V3<T> u{U0.x - U1.x, U0.y - U1.y, U0.z - U1.z};
        V3<T> Ru, D0, D1, D2; arap_rot(ang, u, Ru, D0, D1, D2);
        const T w = A.w_reg;
        const V3<T> res{w * ((O0.x - O1.x) - Ru.x), w * ((O0.y - O1.y) - Ru.y), w * ((O0.z - O1.z) - Ru.z)};
        if (MODE == 0) { if (ok) acc += (double)(T(0.5) * dot3(res, res)); }
        else if (MODE == 1 || MODE == 3) {
            const V3<T> p0 = ld3(v, a0), p1 = ld3(v, a1), pa = ld3(v + offA, a0);
            // J p = w (p0 - p1) - w (D0 pa.x + D1 pa.y + D2 pa.z)
            const V3<T> jp{w * (p0.x - p1.x) - w * (D0.x * pa.x + D1.x * pa.y + D2.x * pa.z),
                           w * (p0.y - p1.y) - w * (D0.y * pa.x + D1.y * pa.y + D2.y * pa.z),
                           w * (p0.z - p1.z) - w * (D0.z * pa.x + D1.z * pa.y + D2.z * pa.z)};
            if (MODE == 1) { const V3<T> m{res.x + jp.x, res.y + jp.y, res.z + jp.z}; if (ok) acc += (double)(T(0.5) * dot3(m, m)); }
            else {
                segmentedAtomicAdd(out, 3 * a0, w * jp.x, ok); segmentedAtomicAdd(out, 3 * a0 + 1, w * jp.y, ok); segmentedAtomicAdd(out, 3 * a0 + 2, w * jp.z, ok);
                segmentedAtomicAdd(out, offA + 3 * a0, -(w * dot3(D0, jp)), ok); segmentedAtomicAdd(out, offA + 3 * a0 + 1, -(w * dot3(D1, jp)), ok);
                segmentedAtomicAdd(out, offA + 3 * a0 + 2, -(w * dot3(D2, jp)), ok);
                if (ok) {
                    plainAtomicAdd(out + 3 * a1, -(w * jp.x)); plainAtomicAdd(out + 3 * a1 + 1, -(w * jp.y)); plainAtomicAdd(out + 3 * a1 + 2, -(w * jp.z));
                    acc += (double)dot3(jp, jp);   // sum_u p_u (J^T J p)_u over this edge's unknowns = |J p|^2 (o.t:2117-2122)
                }
            }
        } else {   // MODE 2: r -= J^T F, diag += J^2
            segmentedAtomicAdd(out, 3 * a0, -(w * res.x), ok); segmentedAtomicAdd(out, 3 * a0 + 1, -(w * res.y), ok); segmentedAtomicAdd(out, 3 * a0 + 2, -(w * res.z), ok);
            segmentedAtomicAdd(out, offA + 3 * a0, w * dot3(D0, res), ok); segmentedAtomicAdd(out, offA + 3 * a0 + 1, w * dot3(D1, res), ok);
            segmentedAtomicAdd(out, offA + 3 * a0 + 2, w * dot3(D2, res), ok);
            const T w2 = w * w;
            segmentedAtomicAdd(out2, 3 * a0, w2, ok); segmentedAtomicAdd(out2, 3 * a0 + 1, w2, ok); segmentedAtomicAdd(out2, 3 * a0 + 2, w2, ok);
            segmentedAtomicAdd(out2, offA + 3 * a0, w2 * dot3(D0, D0), ok); segmentedAtomicAdd(out2, offA + 3 * a0 + 1, w2 * dot3(D1, D1), ok);
            segmentedAtomicAdd(out2, offA + 3 * a0 + 2, w2 * dot3(D2, D2), ok);
            if (ok) {
                plainAtomicAdd(out + 3 * a1, w * res.x); plainAtomicAdd(out + 3 * a1 + 1, w * res.y); plainAtomicAdd(out + 3 * a1 + 2, w * res.z);
                plainAtomicAdd(out2 + 3 * a1, w2); plainAtomicAdd(out2 + 3 * a1 + 1, w2); plainAtomicAdd(out2 + 3 * a1 + 2, w2);
            }
        }
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
}

template <class T>
struct ArapOps : EnergyOps<T> {
    ArapArgs<T> A{};
    int cus = 256;
    ArapOps(const unsigned* dims) {
        A.N = dims[0];
        this->usePreconditioner = true; this->usesGraph = true;                  // arap_mesh_deformation.t:9
        this->addUnknown(2, A.N, 3); this->addUnknown(3, A.N, 3);                // Offset, Angle (:4-5)
        int dev = 0; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    void bind(void** p, LaunchCtx&) override {
        A.w_fit = (T) * (const float*)p[0]; A.w_reg = (T) * (const float*)p[1];
        A.Offset = (const T*)p[2]; A.Angle = (const T*)p[3]; A.UrShape = (const T*)p[4]; A.Constraints = (const T*)p[5];
        A.nE = *(const int*)p[6]; A.v0 = (const int*)p[7]; A.v1 = (const int*)p[8];   // Graph("G", 6, "v0", {N}, 7, "v1", {N}, 8)
    }
    T* unknownPtr(int img) const override { return const_cast<T*>(img == 0 ? A.Offset : A.Angle); }
    int vgrid() const { return (int)std::max<long>(1, std::min<long>((A.N + kBlock - 1) / kBlock, kMaxPartials / 2)); }
    void evalCost(Reduction& out, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(A.nE, cus);
        { ScopedKernel k(ctx, "computeCost"); arap_vertices<T, 0><<<gv, kBlock, 0, ctx.stream>>>(A, nullptr, nullptr, nullptr, nullptr, out.partials); }
        { ScopedKernel k(ctx, "computeCost_Graph"); arap_edges<T, 0><<<ge, kBlock, 0, ctx.stream>>>(A, nullptr, nullptr, nullptr, out.partials + gv); }
        out.n = gv + ge;
    }
    void evalJTF(T* r, T* diag, LaunchCtx& ctx) override {
        { ScopedKernel k(ctx, "PCGInit1"); arap_vertices<T, 2><<<vgrid(), kBlock, 0, ctx.stream>>>(A, nullptr, r, diag, nullptr, nullptr); }
        { ScopedKernel k(ctx, "PCGInit1_Graph"); arap_edges<T, 2><<<edgeGrid(A.nE, cus), kBlock, 0, ctx.stream>>>(A, nullptr, r, diag, nullptr); }
    }
    void applyJTJ(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(A.nE, cus);
        { ScopedKernel k(ctx, "PCGStep1"); arap_vertices<T, 3><<<gv, kBlock, 0, ctx.stream>>>(A, v, out, nullptr, CtC, dot ? dot->partials : nullptr); }
        { ScopedKernel k(ctx, "PCGStep1_Graph"); arap_edges<T, 3><<<ge, kBlock, 0, ctx.stream>>>(A, v, out, nullptr, dot ? dot->partials + gv : nullptr); }
        if (dot) dot->n = gv + ge;
    }
    void evalModelCost(const T* delta, Reduction& out, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(A.nE, cus);
        { ScopedKernel k(ctx, "computeModelCost"); arap_vertices<T, 1><<<gv, kBlock, 0, ctx.stream>>>(A, delta, nullptr, nullptr, nullptr, out.partials); }
        { ScopedKernel k(ctx, "computeModelCost_Graph"); arap_edges<T, 1><<<ge, kBlock, 0, ctx.stream>>>(A, delta, nullptr, nullptr, out.partials + gv); }
        out.n = gv + ge;
    }
};

template <class T> EnergyOps<T>* makeCF(const unsigned* dims) { return new CurveFittingOps<T>(dims); }
template <class T> EnergyOps<T>* makeArap(const unsigned* dims) { return new ArapOps<T>(dims); }

}  // namespace

EnergyInfo curveFittingInfo() {
    EnergyInfo e;
    e.name = "curveFitting"; e.nDims = 2; e.usePreconditioner = true; e.floatOnly = false;
    e.params = {{ParamDecl::kUnknown, "funcParams", "opt_float2", 0}, {ParamDecl::kArray, "data", "opt_float2", 1},
                {ParamDecl::kGraphCount, "G", "int", 2}, {ParamDecl::kGraphIndex, "G.d", "int", 3}, {ParamDecl::kGraphIndex, "G.p", "int", 4}};
    e.makeFloat = makeCF<float>; e.makeDouble = makeCF<double>;
    return e;
}
EnergyInfo arapInfo() {
    EnergyInfo e;
    e.name = "arap_mesh_deformation"; e.nDims = 1; e.usePreconditioner = true; e.floatOnly = false;
    e.params = {{ParamDecl::kScalar, "w_fitSqrt", "float", 0}, {ParamDecl::kScalar, "w_regSqrt", "float", 1},
                {ParamDecl::kUnknown, "Offset", "opt_float3", 2}, {ParamDecl::kUnknown, "Angle", "opt_float3", 3},
                {ParamDecl::kArray, "UrShape", "opt_float3", 4}, {ParamDecl::kArray, "Constraints", "opt_float3", 5},
                {ParamDecl::kGraphCount, "G", "int", 6}, {ParamDecl::kGraphIndex, "G.v0", "int", 7}, {ParamDecl::kGraphIndex, "G.v1", "int", 8}};
    e.makeFloat = makeArap<float>; e.makeDouble = makeArap<double>;
    return e;
}

}  // namespace optamd
