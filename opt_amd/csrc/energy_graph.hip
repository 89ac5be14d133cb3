// Graph-domain energies: arap_mesh_deformation (BASELINE config 4) and curveFitting (the reference's
// tests/minimal_graph_only known-answer test).
//
// Reference behaviour (API/src/o.t:2092-2126, 2228-2253; solverGPUGaussNewton.t:687-706): one thread per
// hyperedge evaluates the residual's Jacobian row block, forms J p, and scatters J^T (J p) into the unknown
// vector with one atomic per (vertex, channel); the per-vertex ("centred") kernel runs first and OVERWRITES,
// the edge kernel then accumulates on top (solver.t:1032-1036, 1062-1065).
//
// MI355X design: no atomics.  The edge lists are turned into per-vertex out- / in-lists once per graph (ensureCsr), and J^T J p is a GATHER:
//   * symmetric graphs whose longest out-list has at most 16 entries (every mesh: OptGraph.h:64-76 emits both directions of every edge; checked per vertex by csr_symmetric):
//     one walk of a vertex's out-list serves both edge directions.  Round 6 layout: what a half-edge needs of its neighbour lives in 16-byte SoA PLANES (D0 / D1: p and its Angle
//     part, rewritten every PCG iteration; T0 / T1: sines / cosines; U0: rest position) and the out-lists in ELL order, one lane per vertex (arap_applyEll) -- the gathers of a
//     wave touch 16 cache lines instead of 64 (the vector L1's line rate bound the 64-byte-record gather of rounds 3-5: profiles/r06_arap_counters.md).  A Gauss-Newton or
//     Levenberg-Marquardt PCG iteration is two kernels -- the flat PCGStep2 + PCGStep3 pass that also rewrites the dynamic planes (arap_flatStepPlanes) and the gather with the
//     sums of the expanded beta numerator; both walk the SAME eighth of the vertices per XCD, the gather forward, the flat pass backward (XcdWalk): 58 -> 47 us per iteration
//     at 500 k vertices, config 4 116.9 -> 95.6 ms;
//   * any other graph: the edge-list gather arap_applyFused (36-byte derivative rows per half-edge) in the reference's three-kernel loop.
// The same inputs give the same bits (fixed summation order per vertex).  The scatter with wave-aggregated atomics (arap_vertices<3> + arap_edges<3>: runs of lanes
// that target the same head vertex folded by a segmented shuffle reduction into one hardware f32 / f64 atomic) is what cost, model cost and curveFitting still use,
// and the J^T J p alternative of development builds (-DOPT_AMD_DEV_SWITCHES, OPT_AMD_ARAP_GATHER=0).
#include "energy.h"
#include "graph_common.h"
#include <hipcub/hipcub.hpp>

namespace optamd {
namespace {

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
    T* D;   // per-edge dR3/da_k (U_v0 - U_v1), k = 0..2: 9 planes of nE scalars, rebuilt by the PCGInit1_Graph pass of every
            // Gauss-Newton iteration so that the PCG loop's edge kernel needs no sin/cos (3 sincos per edge per iteration otherwise)
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
        V3<T> Ru{0, 0, 0}, D0, D1, D2;
        const long ee = ok ? e : 0, nE = A.nE;
        if (MODE == 1 || MODE == 3) {   // PCG loop / model cost: tabulated derivative columns
            D0 = V3<T>{A.D[ee], A.D[nE + ee], A.D[2 * nE + ee]}; D1 = V3<T>{A.D[3 * nE + ee], A.D[4 * nE + ee], A.D[5 * nE + ee]};
            D2 = V3<T>{A.D[6 * nE + ee], A.D[7 * nE + ee], A.D[8 * nE + ee]};
            if (MODE == 1) { V3<T> d0, d1, d2; arap_rot(ang, u, Ru, d0, d1, d2); }
        } else {
            arap_rot(ang, u, Ru, D0, D1, D2);
            if (MODE == 2 && ok) {
                A.D[e] = D0.x; A.D[nE + e] = D0.y; A.D[2 * nE + e] = D0.z; A.D[3 * nE + e] = D1.x; A.D[4 * nE + e] = D1.y; A.D[5 * nE + e] = D1.z;
                A.D[6 * nE + e] = D2.x; A.D[7 * nE + e] = D2.y; A.D[8 * nE + e] = D2.z;
            }
        }
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

// ---- atomics-free J^T J p for ARAP: edge pass + vertex gather ----------------------------------------------------
// The scatter formulation spends its time in 9 atomics per half-edge (measured: 261 us per applyJTJ at 500 k vertices /
// 3 M half-edges, unchanged when the sin/cos were tabulated).  With the edge lists of every vertex known -- the
// half-edges it heads (out) and the ones it tails (in), built once per graph by a counting sort -- the same sums are
// gathers:  (J^T J p)_O(v) = w sum_{e in out(v)} Jp_e - w sum_{e in in(v)} Jp_e,   (J^T J p)_a(v)_k = -w sum_{e in out(v)} D_{e,k} . Jp_e.
// Pass 1 writes Jp_e (3 scalars per half-edge), pass 2 is one thread per vertex.  Deterministic (lists are sorted by
// edge id), no atomics, and the per-vertex kernel of the scatter path is folded into pass 2.
struct GraphCsr { const int* outOff; const int* outIdx; const int* inOff; const int* inIdx; const int* outNbr = nullptr; const int* inNbr = nullptr; };
// neighbour vertex of every list slot (tail of an out-edge, head of an in-edge): saves the fused kernel one level of dependent gathers
__global__ __launch_bounds__(kBlock) void csr_neighbours(const int* __restrict__ v0, const int* __restrict__ v1, int nE, const int* __restrict__ outIdx, const int* __restrict__ inIdx,
                                                         int* __restrict__ outNbr, int* __restrict__ inNbr) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nE; k += gridDim.x * blockDim.x) { outNbr[k] = v1[outIdx[k]]; inNbr[k] = v0[inIdx[k]]; }
}

__global__ __launch_bounds__(kBlock) void csr_count(const int* __restrict__ v0, const int* __restrict__ v1, int nE, int* __restrict__ outDeg, int* __restrict__ inDeg) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nE; e += gridDim.x * blockDim.x) { atomicAdd(outDeg + v0[e], 1); atomicAdd(inDeg + v1[e], 1); }
}
__global__ __launch_bounds__(kBlock) void csr_fill(const int* __restrict__ v0, const int* __restrict__ v1, int nE, const int* __restrict__ outOff, const int* __restrict__ inOff,
                                                   int* __restrict__ outCur, int* __restrict__ inCur, int* __restrict__ outIdx, int* __restrict__ inIdx) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nE; e += gridDim.x * blockDim.x) {
        const int a = v0[e], b = v1[e];
        outIdx[outOff[a] + atomicAdd(outCur + a, 1)] = e;
        inIdx[inOff[b] + atomicAdd(inCur + b, 1)] = e;
    }
}
__global__ __launch_bounds__(kBlock) void csr_sort(long N, const int* __restrict__ off, int* __restrict__ idx) {   // per-vertex insertion sort: fixed summation order
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < N; v += (long)gridDim.x * blockDim.x) {
        const int b = off[v], e = off[v + 1];
        for (int i = b + 1; i < e; ++i) { const int x = idx[i]; int j = i - 1; while (j >= b && idx[j] > x) { idx[j + 1] = idx[j]; --j; } idx[j + 1] = x; }
    }
}
__global__ __launch_bounds__(kBlock) void csr_checksum(const int* __restrict__ v0, const int* __restrict__ v1, int nE, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nE; e += gridDim.x * blockDim.x)
        acc += ((unsigned long long)(unsigned)v0[e] * 0x9E3779B97F4A7C15ull + (unsigned long long)(unsigned)v1[e] * 0xC2B2AE3D27D4EB4Full) ^ ((unsigned long long)e * 0x165667B19E3779F9ull);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
    __shared__ unsigned long long part[kBlock / kWave];      // one atomic per workgroup: 12 000 wave-level atomics on one address took 100 us per bind
    if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int w = 0; w < kBlock / kWave; ++w) t += part[w]; atomicAdd(out, t); }
}

// 8 lanes per vertex, each walking one slot of the out-list and one of the in-list (a mesh vertex has ~6 of each), so the
// dependent index -> record gathers of one vertex are in flight together instead of in a 12-trip serial loop.  The gather kernels are
// bound by how many such chains are in flight (time ~ 1 / resident workgroups), hence the full-occupancy grid.
#ifndef ARAP_LANES
#define ARAP_LANES 8
#endif
constexpr int kLanesPerVertex = ARAP_LANES;

// ---- edge pass and vertex gather in ONE launch (round 2) --------------------------------------------------------------------------
// The two-pass form moves every half-edge's 24-byte record through memory twice more than needed: written by the edge pass, read back as
// an out-record by its head vertex and as an in-record by its tail vertex (edge pass 240 MB + vertex pass 279 MB per J^T J p at 500 k
// vertices / 3 M half-edges).  Here the lane that would read a record computes it instead: for its out-edge e = (v -> u) from p_u and the
// edge's derivative columns D_e, for its in-edge e' = (u' -> v) from p_u', pa_u' and D_e' -- the per-edge expressions of arap_edges<3>, summed per vertex in a fixed lane / shuffle order.  D is kept as 36-byte AoS rows (D9) for this kernel: a gathered in-edge costs one or two cache
// lines instead of nine plane accesses.  Neighbour p's are gathers that mostly hit L2 (a mesh neighbour is a memory neighbour).
template <class T>
__global__ __launch_bounds__(kBlock) void arap_packD(const T* __restrict__ D, T* __restrict__ D9, long nE) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < 9 * nE; i += (long)gridDim.x * blockDim.x) D9[i] = D[(i % 9) * nE + i / 9];
}
// S.r != nullptr (round 3): the launch is the Step1 half of a TWO-kernel PCG iteration (arap_flatStepPlanes + arap_applyEll; EnergyOps::pcgIteration).  Besides p . A p it then sums, at the
// lane that owns the vertex, what the expanded beta numerator of the next flat pass needs -- sum M r^2, sum M r . A p, sum M (A p)^2, every term from M, r, A p themselves in double
// (exact products of floats; see energy_image_warping.hip dprod3) -- so that PCGStep2 and PCGStep3 become one flat pass without a reduction between them.
template <class T> __device__ __forceinline__ double arap_dprod3(T m, T a, T b) { return ((double)m * (double)a) * (double)b; }
struct ArapIterSums { const void* r; const void* M; double *aNum, *s2, *s3; };
template <class T>
__global__ __launch_bounds__(kBlock) void arap_applyFused(ArapArgs<T> A, GraphCsr G, const T* __restrict__ D9, const T* __restrict__ v, T* __restrict__ out,
                                                          const T* __restrict__ CtC, double* __restrict__ partials, ArapIterSums S = ArapIterSums{nullptr, nullptr, nullptr, nullptr, nullptr}) {
    __shared__ double scratch[4 * (kBlock / kWave + 1)];
    double acc = 0, accNum = 0, acc2 = 0, acc3 = 0;
    const T* rv = (const T*)S.r; const T* Mv = (const T*)S.M;
    const long offA = 3 * A.N;
    const int sub = threadIdx.x % kLanesPerVertex, slot = sub;
    const long nGroups = (A.N + (kBlock / kLanesPerVertex) - 1) / (kBlock / kLanesPerVertex);
    for (long g = blockIdx.x; g < nGroups; g += gridDim.x) {
        const long i = g * (kBlock / kLanesPerVertex) + threadIdx.x / kLanesPerVertex;
        const bool ok = i < A.N;
        const long iv = ok ? i : 0;
        const T w = A.w_reg;
        const V3<T> pv = ld3(v, iv), pav = ld3(v + offA, iv);
        V3<T> rO{0, 0, 0}, rA{0, 0, 0}, mO{0, 0, 0}, mA{0, 0, 0};
        if (rv) { rO = ld3(rv, iv); rA = ld3(rv + offA, iv); mO = ld3(Mv, iv); mA = ld3(Mv + offA, iv); }      // requested before the edge walk: known from the vertex index alone
        T s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
        const int bo = G.outOff[iv], eo = ok ? G.outOff[iv + 1] : bo, bi = G.inOff[iv], ei = ok ? G.inOff[iv + 1] : bi;
        for (int k = 0; k < max(eo - bo, ei - bi); k += kLanesPerVertex) {
            const int ko = bo + slot + k, ki = bi + slot + k;
            const int eOut = ko < eo ? G.outIdx[ko] : -1, eIn = ki < ei ? G.inIdx[ki] : -1;
            const long e = max(eOut, 0), f = max(eIn, 0);
            const long u = ko < eo ? G.outNbr[ko] : 0, uIn = ki < ei ? G.inNbr[ki] : 0;      // tail of my out-edge, head of my in-edge (csr_neighbours)
            const V3<T> pu = ld3(v, u), pq = ld3(v, uIn), paq = ld3(v + offA, uIn);
            const T* d = D9 + 9 * e; const T* h = D9 + 9 * f;
            const T wo = eOut >= 0 ? w : T(0), wi = eIn >= 0 ? w : T(0);
            {   // out-edge (v -> u): J p and D_k . J p  (arap_edges<3> with v0 = v)
                const T jx = w * (pv.x - pu.x) - w * (d[0] * pav.x + d[3] * pav.y + d[6] * pav.z);
                const T jy = w * (pv.y - pu.y) - w * (d[1] * pav.x + d[4] * pav.y + d[7] * pav.z);
                const T jz = w * (pv.z - pu.z) - w * (d[2] * pav.x + d[5] * pav.y + d[8] * pav.z);
                s0 += wo * jx; s1 += wo * jy; s2 += wo * jz;
                s3 -= wo * (d[0] * jx + d[1] * jy + d[2] * jz); s4 -= wo * (d[3] * jx + d[4] * jy + d[5] * jz); s5 -= wo * (d[6] * jx + d[7] * jy + d[8] * jz);
                if (eOut >= 0) acc += (double)(jx * jx + jy * jy + jz * jz);              // sum_u p_u (J^T J p)_u of this edge = |J p|^2 (o.t:2117-2122)
            }
            {   // in-edge (u' -> v): only its J p reaches this vertex's Offset row
                const T jx = w * (pq.x - pv.x) - w * (h[0] * paq.x + h[3] * paq.y + h[6] * paq.z);
                const T jy = w * (pq.y - pv.y) - w * (h[1] * paq.x + h[4] * paq.y + h[7] * paq.z);
                const T jz = w * (pq.z - pv.z) - w * (h[2] * paq.x + h[5] * paq.y + h[8] * paq.z);
                s0 -= wi * jx; s1 -= wi * jy; s2 -= wi * jz;
            }
        }
#pragma unroll
        for (int m = 1; m < kLanesPerVertex; m <<= 1) {
            s0 += __shfl_xor(s0, m, kWave); s1 += __shfl_xor(s1, m, kWave); s2 += __shfl_xor(s2, m, kWave);
            s3 += __shfl_xor(s3, m, kWave); s4 += __shfl_xor(s4, m, kWave); s5 += __shfl_xor(s5, m, kWave);
        }
        if (ok && sub == 0) {
            const bool valid = A.Constraints[3 * i] >= T(-999999.9);
            const T wf = valid ? A.w_fit : T(0);
            V3<T> q{wf * wf * pv.x, wf * wf * pv.y, wf * wf * pv.z}, qa{0, 0, 0};
            if (CtC) { const V3<T> cO = ld3(CtC, i), cA = ld3(CtC + offA, i); q.x += cO.x * pv.x; q.y += cO.y * pv.y; q.z += cO.z * pv.z; qa.x = cA.x * pav.x; qa.y = cA.y * pav.y; qa.z = cA.z * pav.z; }
            acc += (double)(dot3(pv, q) + dot3(pav, qa));
            const V3<T> oO{q.x + s0, q.y + s1, q.z + s2}, oA{qa.x + s3, qa.y + s4, qa.z + s5};
            out[3 * i] = oO.x; out[3 * i + 1] = oO.y; out[3 * i + 2] = oO.z;
            out[offA + 3 * i] = oA.x; out[offA + 3 * i + 1] = oA.y; out[offA + 3 * i + 2] = oA.z;
            if (rv) {
                accNum += arap_dprod3(mO.x, rO.x, rO.x) + arap_dprod3(mO.y, rO.y, rO.y) + arap_dprod3(mO.z, rO.z, rO.z) + arap_dprod3(mA.x, rA.x, rA.x) + arap_dprod3(mA.y, rA.y, rA.y) + arap_dprod3(mA.z, rA.z, rA.z);
                acc2 += arap_dprod3(mO.x, rO.x, oO.x) + arap_dprod3(mO.y, rO.y, oO.y) + arap_dprod3(mO.z, rO.z, oO.z) + arap_dprod3(mA.x, rA.x, oA.x) + arap_dprod3(mA.y, rA.y, oA.y) + arap_dprod3(mA.z, rA.z, oA.z);
                acc3 += arap_dprod3(mO.x, oO.x, oO.x) + arap_dprod3(mO.y, oO.y, oO.y) + arap_dprod3(mO.z, oO.z, oO.z) + arap_dprod3(mA.x, oA.x, oA.x) + arap_dprod3(mA.y, oA.y, oA.y) + arap_dprod3(mA.z, oA.z, oA.z);
            }
        }
    }
    if (rv) {
        double vv[4] = {acc, accNum, acc2, acc3};
        blockReduceSumN<4>(vv, scratch);
        if (threadIdx.x == 0) { if (partials) partials[blockIdx.x] = vv[0]; S.aNum[blockIdx.x] = vv[1]; S.s2[blockIdx.x] = vv[2]; S.s3[blockIdx.x] = vv[3]; }
        return;
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
}

// PCGStep2 + PCGStep3 of iteration k-1 as ONE flat pass (solver.t:446-489, 537-550): alpha from the previous gather's sums, beta from their expansion
//   sum M (r - alpha A p)^2 = aNum - 2 alpha s2 + alpha^2 s3   (all in double, clamped at 0 like the direct sum it replaces; energy.h PcgIterArgs),
// then delta += alpha p, r -= alpha A p, z = M r, p = z + beta p -- no reduction in this kernel, none between Step2 and Step3.
// ---- round 3: J^T J p on a SYMMETRIC graph from one record per vertex ---------------------------------------------------------------------------
// arap_applyFused is bound by the number of dependent index -> record chains in flight: per half-edge slot it follows outIdx / inIdx / outNbr / inNbr (16 B), two 36-byte
// derivative rows and three 12-byte vectors of the neighbour, 3-5 cache lines.  A mesh carries every edge in both directions (examples/shared/OptGraph.h builds v0 -> v1 for
// every neighbour of every vertex), so the in-edges of v are the reversed out-edges of v: (u -> v) exists exactly as often as (v -> u) -- checked per vertex when the edge
// lists are built (csr_symmetric), any other graph keeps arap_applyFused.  Then one walk of the out-list serves both directions, and everything the walk needs from a
// neighbour u -- p_u, pa_u and the sines / cosines of its angles, from which its derivative columns D_k(u -> v) = dR3/da_k(a_u) (U_u - U_v) follow in ~60 flops instead of a
// 36-byte row -- is ONE 64-byte record (one cache line); the slot itself is 16 contiguous bytes {U_v - U_u, u}.  Per half-edge pair: 80 B in two lines instead of ~190 B in
// seven.  The expressions are arap_edges<3>'s / arap_rot's (same association: J p = w (p_v0 - p_v1) - w (D_0 pa.x + D_1 pa.y + D_2 pa.z)), the in-edge terms are summed in
// out-list order.
// ---- symmetric-graph path, round 6 layout: 16-byte SoA PLANES + ELL out-lists, one lane per vertex ---------------------------------------------------------
// What a half-edge's arithmetic needs of its neighbour u -- p_u, the Angle part of p_u, the sines / cosines of u's angles, u's rest position -- used to be one 64-byte
// record per vertex, gathered with 16-byte loads: every lane of a gather touched a cache line of its own, four times over, and the vector L1's line rate (one line per clock
// per CU: 10.9 M line accesses = 17.8 us of a 29 us launch) bound the kernel together with its issue slots (profiles/r06_arap_counters.md).  Now every group of four scalars
// is a PLANE of N 16-byte entries: the neighbours of consecutive vertices of a mesh are consecutive vertices, so the 64 lanes of a gather touch 16 lines instead of 64.
//   D0 = {p.x, p.y, p.z, pa.x}   D1 = {pa.y, pa.z, -, -}      dynamic: rewritten by the flat pass of every PCG iteration (two coalesced 16-byte stores per vertex)
//   T0 = {sa, ca, sb, cb}        T1 = {sg, cg, -, -}           per Gauss-Newton step (arap_buildStatic)
//   U0 = {U.x, U.y, U.z, -}                                     rest positions (rebuilt with T0 / T1: UrShape is an input of the solve)
// and the out-lists are stored ELL-wise: ell[j * N + v] = the j-th out-neighbour of v (v itself beyond its degree, weight 0), so slot j of consecutive vertices is contiguous.
// A vertex with more than kEllMax neighbours sends the graph to the edge-list gather (arap_applyFused), like an asymmetric one.
constexpr int kEllMax = 16;
template <class T> struct alignas(16) Q4 { T a, b, c, d; };
template <class T> struct ArapPlanes { Q4<T>* D0; Q4<T>* D1; Q4<T>* T0; Q4<T>* T1; Q4<T>* U0; const int* ell; const int* deg; int K; };
template <class T> struct ArapCoef { T a0, a1, a2, a3, a4, a5, b0, b1, b2, b3, b4, b5, b6, b7, b8, c0, c1, c2, c3, c4, c5; };
// the non-zero entries of dR3/d alpha, d beta, d gamma (arap_rot's coefficients, same products)
template <class T>
__device__ __forceinline__ ArapCoef<T> arap_coef(T sa, T ca, T sb, T cb, T sg, T cg) {
    ArapCoef<T> c;
    c.a0 = sg * sa + cg * sb * ca;  c.a1 = sg * ca - cg * sb * sa;
    c.a2 = -cg * sa + sg * sb * ca; c.a3 = -cg * ca - sg * sb * sa;
    c.a4 = cb * ca;                 c.a5 = -cb * sa;
    c.b0 = -cg * sb; c.b1 = cg * cb * sa; c.b2 = cg * cb * ca;
    c.b3 = -sg * sb; c.b4 = sg * cb * sa; c.b5 = sg * cb * ca;
    c.b6 = -cb;      c.b7 = -sb * sa;     c.b8 = -sb * ca;
    c.c0 = -sg * cb; c.c1 = -cg * ca - sg * sb * sa; c.c2 = cg * sa - sg * sb * ca;
    c.c3 = cg * cb;  c.c4 = -sg * ca + cg * sb * sa; c.c5 = sg * sa + cg * sb * ca;
    return c;
}
template <class T>
__device__ __forceinline__ void arap_cols(const ArapCoef<T>& c, const V3<T>& u, V3<T>& D0, V3<T>& D1, V3<T>& D2) {
    D0.x = c.a0 * u.y + c.a1 * u.z; D0.y = c.a2 * u.y + c.a3 * u.z; D0.z = c.a4 * u.y + c.a5 * u.z;
    D1.x = c.b0 * u.x + c.b1 * u.y + c.b2 * u.z; D1.y = c.b3 * u.x + c.b4 * u.y + c.b5 * u.z; D1.z = c.b6 * u.x + c.b7 * u.y + c.b8 * u.z;
    D2.x = c.c0 * u.x + c.c1 * u.y + c.c2 * u.z; D2.y = c.c3 * u.x + c.c4 * u.y + c.c5 * u.z; D2.z = 0;
}
template <class T> __device__ __forceinline__ V3<T> ldv3(const T* p, long i) { return reinterpret_cast<const V3<T>*>(p)[i]; }      // one 12 / 24-byte load
template <class T> __device__ __forceinline__ void stv3(T* p, long i, const V3<T>& v) { reinterpret_cast<V3<T>*>(p)[i] = v; }
// per vertex: is the multiset of out-neighbours the multiset of in-neighbours?  (quadratic in the degree; lists longer than 64 are declared asymmetric)
__global__ __launch_bounds__(kBlock) void csr_symmetric(long N, const int* __restrict__ outOff, const int* __restrict__ outNbr, const int* __restrict__ inOff, const int* __restrict__ inNbr,
                                                        int* __restrict__ notSym) {
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < N; v += (long)gridDim.x * blockDim.x) {
        const int bo = outOff[v], eo = outOff[v + 1], bi = inOff[v], ei = inOff[v + 1];
        bool bad = (eo - bo) != (ei - bi) || (eo - bo) > 64;
        for (int j = bo; j < eo && !bad; ++j) {
            const int x = outNbr[j];
            int co = 0, ci = 0;
            for (int k = bo; k < eo; ++k) co += outNbr[k] == x;
            for (int k = bi; k < ei; ++k) ci += inNbr[k] == x;
            bad = co != ci;
        }
        if (bad) *notSym = 1;
    }
}
// out-lists in ELL order (once per graph): ell[j * N + v], padded with v itself; deg[v]; *maxDeg (a vertex beyond kEllMax sends the graph to the edge-list gather)
__global__ __launch_bounds__(kBlock) void arap_buildEll(long N, const int* __restrict__ outOff, const int* __restrict__ outNbr, int K, int* __restrict__ ell, int* __restrict__ deg) {
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < N; v += (long)gridDim.x * blockDim.x) {
        const int b = outOff[v], d = outOff[v + 1] - b;
        deg[v] = d;
        for (int j = 0; j < K; ++j) ell[(long)j * N + v] = j < d ? outNbr[b + j] : (int)v;
    }
}
__global__ __launch_bounds__(kBlock) void csr_maxDegree(long N, const int* __restrict__ outOff, int* __restrict__ out) {
    int m = 0;
    for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < N; v += (long)gridDim.x * blockDim.x) m = max(m, outOff[v + 1] - outOff[v]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_down(m, o, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(out, m);
}
// once per Gauss-Newton iteration: the sines and cosines of every vertex's angles and its rest position
template <class T>
__global__ __launch_bounds__(kBlock) void arap_buildStatic(ArapArgs<T> A, ArapPlanes<T> P) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < A.N; i += (long)gridDim.x * blockDim.x) {
        const V3<T> a = ldv3(A.Angle, i), U = ldv3(A.UrShape, i);
        T sa, ca, sb, cb, sg, cg;
        sincosT(a.x, &sa, &ca); sincosT(a.y, &sb, &cb); sincosT(a.z, &sg, &cg);
        P.T0[i] = Q4<T>{sa, ca, sb, cb}; P.T1[i] = Q4<T>{sg, cg, T(0), T(0)}; P.U0[i] = Q4<T>{U.x, U.y, U.z, T(0)};
    }
}
// the dynamic planes from a solver vector (the first launch of a loop, probes, the A delta of LM's residual reset)
template <class T>
__global__ __launch_bounds__(kBlock) void arap_packPlanes(const T* __restrict__ v, ArapPlanes<T> P, long N) {
    const long offA = 3 * N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const V3<T> p = ldv3(v, i), pa = ldv3(v + offA, i);
        P.D0[i] = Q4<T>{p.x, p.y, p.z, pa.x}; P.D1[i] = Q4<T>{pa.y, pa.z, T(0), T(0)};
    }
}
// PCGStep3 (k_step3 in solver.hip: solver.t:537-550) writing the new search direction to the solver's vector AND to the dynamic planes (the reference-ordered loop and the
// generic Levenberg-Marquardt loop: EnergyOps::applyJTJFused)
template <class T>
__global__ __launch_bounds__(kBlock) void arap_step3Planes(const T* __restrict__ z, const T* __restrict__ pOld, T* __restrict__ pNew, ArapPlanes<T> P, long N,
                                                           const double* __restrict__ bNumPartials, int nB, const double* __restrict__ aNumOld, double* __restrict__ aNumNext) {
    __shared__ double scratch[kBlock / kWave + 1];
    const double bSum = sumPartials(bNumPartials, nB, scratch);
    const T rDotzNew = (T)bSum, rDotzOld = (T)aNumOld[0];
    const T beta = (rDotzOld > T(0)) ? rDotzNew / rDotzOld : T(0);
    if (blockIdx.x == 0 && threadIdx.x == 0) aNumNext[0] = bSum;
    const long offA = 3 * N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const V3<T> zo = ldv3(z, i), za = ldv3(z + offA, i), po = ldv3(pOld, i), pa = ldv3(pOld + offA, i);
        const V3<T> no{zo.x + beta * po.x, zo.y + beta * po.y, zo.z + beta * po.z}, na{za.x + beta * pa.x, za.y + beta * pa.y, za.z + beta * pa.z};
        stv3(pNew, i, no); stv3(pNew + offA, i, na);
        P.D0[i] = Q4<T>{no.x, no.y, no.z, na.x}; P.D1[i] = Q4<T>{na.y, na.z, T(0), T(0)};
    }
}
// XCD-aware, cache-friendly order of the two kernels of an iteration.  Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MB L2, and a vertex's plane entries
// are read by the vertex and by its ~6 neighbours -- which in a mesh's vertex order sit a row of vertices apart: XCD x walks the x-th eighth of the vertex groups, so the copies
// meet in one L2 (xcd != 0; the grid is a multiple of 8).  The gather walks its eighth FORWARD, the flat pass that follows it BACKWARD over the SAME eighth: what one kernel
// touched last (the gather's A p, the flat pass's planes and vectors) is what the next one touches first, while it is still in that XCD's L2 / the Infinity Cache.
struct XcdWalk {
    long first, step, count;
    __device__ __forceinline__ XcdWalk(long nGroups, int xcd) {
        const long perXcd = (nGroups + 7) / 8, wgPerXcd = gridDim.x / 8;
        first = xcd ? (long)(blockIdx.x % 8) * perXcd + blockIdx.x / 8 : blockIdx.x;
        step = xcd ? wgPerXcd : gridDim.x;
        const long end = xcd ? min(nGroups, (long)(blockIdx.x % 8 + 1) * perXcd) : nGroups;
        count = end > first ? (end - first + step - 1) / step : 0;
    }
    __device__ __forceinline__ long group(long t, bool backward) const { return first + (backward ? count - 1 - t : t) * step; }
};
// J^T J p as a gather over the vertex's out-list (symmetric graphs), ONE lane per vertex.  BATCH neighbours of a lane are in flight together: their ids (coalesced: ELL),
// then their five plane entries each, then the arithmetic -- both edge directions of a half-edge pair from one walk, as arap_edges<3> would compute them.
#ifndef ARAP_ELL_BATCH
#define ARAP_ELL_BATCH 3
#endif
template <class T, int BATCH>
__global__ __launch_bounds__(kBlock) void arap_applyEll(ArapArgs<T> A, ArapPlanes<T> P, const T* __restrict__ v, T* __restrict__ out, const T* __restrict__ CtC, double* __restrict__ partials,
                                                        int xcd, ArapIterSums S) {
    __shared__ double scratch[4 * (kBlock / kWave + 1)];
    double acc = 0, accNum = 0, acc2 = 0, acc3 = 0;      // S.r != nullptr: the Step1 half of a two-kernel PCG iteration, see arap_applyFused
    const T* rv = (const T*)S.r; const T* Mv = (const T*)S.M;
    const long N = A.N, offA = 3 * N;
    const XcdWalk walk((N + kBlock - 1) / kBlock, xcd);
    const T w = A.w_reg;
    for (long t = 0; t < walk.count; ++t) {       // (trip counts differ between workgroups only: the wave-wide maximum below needs whole waves, not whole grids)
        const long g = walk.group(t, false);
        const long i = g * kBlock + threadIdx.x;
        const bool ok = i < N;
        const long iv = ok ? i : 0;
        const Q4<T> d0 = P.D0[iv], d1 = P.D1[iv], t0 = P.T0[iv], t1 = P.T1[iv], u0 = P.U0[iv];
        const int deg = ok ? P.deg[iv] : 0;
        int dmax = deg;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, __shfl_xor(dmax, o, kWave));      // the wave walks as many slots as its longest list
        V3<T> rO{0, 0, 0}, rA{0, 0, 0}, mO{0, 0, 0}, mA{0, 0, 0};
        if (rv) { rO = ldv3(rv, iv); rA = ldv3(rv + offA, iv); mO = ldv3(Mv, iv); mA = ldv3(Mv + offA, iv); }      // requested before the walk: known from the vertex index alone
        const V3<T> pv{d0.a, d0.b, d0.c}, pav{d0.d, d1.a, d1.b};
        const ArapCoef<T> cv = arap_coef(t0.a, t0.b, t0.c, t0.d, t1.a, t1.b);
        T s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
        for (int j0 = 0; j0 < dmax; j0 += BATCH) {
            int nid[BATCH]; T wm[BATCH];
            Q4<T> nd0[BATCH], nd1[BATCH], nt0[BATCH], nt1[BATCH], nu0[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) { const int jj = j0 + j; wm[j] = jj < deg ? w : T(0); nid[j] = P.ell[(long)min(jj, P.K - 1) * N + iv]; }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) { nd0[j] = P.D0[nid[j]]; nd1[j] = P.D1[nid[j]]; nt0[j] = P.T0[nid[j]]; nt1[j] = P.T1[nid[j]]; nu0[j] = P.U0[nid[j]]; }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const T wj = wm[j];
                const V3<T> np{nd0[j].a, nd0[j].b, nd0[j].c}, npa{nd0[j].d, nd1[j].a, nd1[j].b};
                const V3<T> u{u0.a - nu0[j].a, u0.b - nu0[j].b, u0.c - nu0[j].c}, un{-u.x, -u.y, -u.z};      // U_v - U_u: the subtraction the slots of rounds 3-5 stored
                V3<T> D0, D1, D2;
                arap_cols(cv, u, D0, D1, D2);
                {   // out-edge (v -> u): J p and D_k . J p  (arap_edges<3> with v0 = v)
                    const T jx = w * (pv.x - np.x) - w * (D0.x * pav.x + D1.x * pav.y + D2.x * pav.z);
                    const T jy = w * (pv.y - np.y) - w * (D0.y * pav.x + D1.y * pav.y + D2.y * pav.z);
                    const T jz = w * (pv.z - np.z) - w * (D0.z * pav.x + D1.z * pav.y + D2.z * pav.z);
                    s0 += wj * jx; s1 += wj * jy; s2 += wj * jz;
                    s3 -= wj * (D0.x * jx + D0.y * jy + D0.z * jz); s4 -= wj * (D1.x * jx + D1.y * jy + D1.z * jz); s5 -= wj * (D2.x * jx + D2.y * jy + D2.z * jz);
                    if (wj != T(0)) acc += (double)(jx * jx + jy * jy + jz * jz);              // sum_u p_u (J^T J p)_u of this edge = |J p|^2 (o.t:2117-2122)
                }
                {   // its reverse (u -> v): only its J p reaches this vertex's Offset row
                    const ArapCoef<T> cu = arap_coef(nt0[j].a, nt0[j].b, nt0[j].c, nt0[j].d, nt1[j].a, nt1[j].b);
                    V3<T> E0, E1, E2;
                    arap_cols(cu, un, E0, E1, E2);
                    const T jx = w * (np.x - pv.x) - w * (E0.x * npa.x + E1.x * npa.y + E2.x * npa.z);
                    const T jy = w * (np.y - pv.y) - w * (E0.y * npa.x + E1.y * npa.y + E2.y * npa.z);
                    const T jz = w * (np.z - pv.z) - w * (E0.z * npa.x + E1.z * npa.y + E2.z * npa.z);
                    s0 -= wj * jx; s1 -= wj * jy; s2 -= wj * jz;
                }
            }
        }
        if (ok) {
            // per-vertex ("centred") part: fitting term and, for LM, CtC p  -- what arap_vertices<3> computes
            const bool valid = A.Constraints[3 * i] >= T(-999999.9);
            const T wf = valid ? A.w_fit : T(0);
            V3<T> q{wf * wf * pv.x, wf * wf * pv.y, wf * wf * pv.z}, qa{0, 0, 0};
            if (CtC) { const V3<T> cO = ldv3(CtC, i), cA = ldv3(CtC + offA, i); q.x += cO.x * pv.x; q.y += cO.y * pv.y; q.z += cO.z * pv.z; qa.x = cA.x * pav.x; qa.y = cA.y * pav.y; qa.z = cA.z * pav.z; }
            acc += (double)(dot3(pv, q) + dot3(pav, qa));
            const V3<T> oO{q.x + s0, q.y + s1, q.z + s2}, oA{qa.x + s3, qa.y + s4, qa.z + s5};
            stv3(out, i, oO); stv3(out + offA, i, oA);
            if (rv) {
                accNum += arap_dprod3(mO.x, rO.x, rO.x) + arap_dprod3(mO.y, rO.y, rO.y) + arap_dprod3(mO.z, rO.z, rO.z) + arap_dprod3(mA.x, rA.x, rA.x) + arap_dprod3(mA.y, rA.y, rA.y) + arap_dprod3(mA.z, rA.z, rA.z);
                acc2 += arap_dprod3(mO.x, rO.x, oO.x) + arap_dprod3(mO.y, rO.y, oO.y) + arap_dprod3(mO.z, rO.z, oO.z) + arap_dprod3(mA.x, rA.x, oA.x) + arap_dprod3(mA.y, rA.y, oA.y) + arap_dprod3(mA.z, rA.z, oA.z);
                acc3 += arap_dprod3(mO.x, oO.x, oO.x) + arap_dprod3(mO.y, oO.y, oO.y) + arap_dprod3(mO.z, oO.z, oO.z) + arap_dprod3(mA.x, oA.x, oA.x) + arap_dprod3(mA.y, oA.y, oA.y) + arap_dprod3(mA.z, oA.z, oA.z);
            }
        }
    }
    if (rv) {
        double vv[4] = {acc, accNum, acc2, acc3};
        blockReduceSumN<4>(vv, scratch);
        if (threadIdx.x == 0) { if (partials) partials[blockIdx.x] = vv[0]; S.aNum[blockIdx.x] = vv[1]; S.s2[blockIdx.x] = vv[2]; S.s3[blockIdx.x] = vv[3]; }
        return;
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
}

// PCGStep2 + PCGStep3 in one flat pass for the plane path, one thread per vertex: 12-byte (24-byte) loads of the vertex's Offset and Angle parts of delta, p, r, A p, M, the
// updates, and the vertex's dynamic planes in two coalesced 16-byte stores (rounds 3-5 moved 16-byte packs and passed the new p through LDS to a thread that wrote 24 B
// of a 64 B record with six dword stores -- 41 % of the launch stalled on issue).
// LM (Levenberg-Marquardt, round 6): the same pass with the reference's LM extras -- delta goes to deltaOut (the solver enqueues the next launch before it has read Q:
// an early-out must still find the old delta), Q_{k-1} = 1/2 sum delta . (r + b) (solver.t:483-485) leaves as per-workgroup partials (tagged words if qTag != 0), and
// after a split residual reset (L.afterReset: delta and r are already the new ones, solver.t:1077-1083) the pass only forms p = M r + beta p with beta = sum bNum / sum bDen.
struct ArapLmStep { const void* b; void* deltaOut; double* q; unsigned qTag; int afterReset; const double* bNumP; int nbNum; const double* bDenP; int nbDen; };
template <class T, bool LM>
__global__ __launch_bounds__(kBlock) void arap_flatStepPlanes(ArapPlanes<T> P, long N, T* __restrict__ delta, const T* __restrict__ pOld, const T* __restrict__ rOld, const T* __restrict__ Ap,
                                                              const T* __restrict__ M, T* __restrict__ rNew, T* __restrict__ pNew,
                                                              const double* aNumP, int nNum, const double* aDenP, int nDen, const double* s2P, int n2, const double* s3P, int n3, ArapLmStep L, int xcd, int backward) {
    __shared__ double scratch[4 * (kBlock / kWave + 1)];
    T alpha, beta;
    const bool restart = LM && L.afterReset;
    if (restart) {
        const double* const ps[2] = {L.bNumP, L.bDenP}; const int ns[2] = {L.nbNum, L.nbDen}; double o2[2];
        sumPartialsN<2>(ps, ns, scratch, o2);
        const T bNum = (T)o2[0], bDen = (T)o2[1];
        alpha = T(0); beta = (bDen > T(0)) ? bNum / bDen : T(0);                            // PCGStep3's guard (solver.t:544-547)
    } else {
        const double* const ps[4] = {aNumP, aDenP, s2P, s3P}; const int ns[4] = {nNum, nDen, n2, n3}; double o4[4];
        sumPartialsN<4>(ps, ns, scratch, o4);
        const T aNum = (T)o4[0], aDen = (T)o4[1];
        alpha = (aDen > T(0)) ? aNum / aDen : T(0);                                          // solver.t:456-459
        const double bNumD = fmax(o4[0] - 2.0 * (double)alpha * o4[2] + (double)alpha * (double)alpha * o4[3], 0.0);
        beta = (aNum > T(0)) ? (T)bNumD / aNum : T(0);                                       // solver.t:544-547
    }
    const T* const bV = (const T*)L.b; T* const dOut = LM ? (T*)L.deltaOut : delta;
    double accQ = 0;
    const long offA = 3 * N;
    const XcdWalk walk((N + kBlock - 1) / kBlock, xcd);
    for (long t = 0; t < walk.count; ++t) {      // the gather's eighths, walked backward (XcdWalk)
        const long i = walk.group(t, backward != 0) * kBlock + threadIdx.x;
        if (i >= N) continue;
        V3<T> pn[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long o = h ? offA : 0;
            const V3<T> p = ldv3(pOld + o, i), r = ldv3(rOld + o, i), m = ldv3(M + o, i);
            V3<T> rn;
            if (restart) {
                rn = r;
                pn[h] = V3<T>{m.x * r.x + beta * p.x, m.y * r.y + beta * p.y, m.z * r.z + beta * p.z};
            } else {
                const V3<T> d0 = ldv3(delta + o, i), a = ldv3(Ap + o, i);
                const V3<T> d{d0.x + alpha * p.x, d0.y + alpha * p.y, d0.z + alpha * p.z};
                rn = V3<T>{r.x - alpha * a.x, r.y - alpha * a.y, r.z - alpha * a.z};
                const V3<T> z{m.x * rn.x, m.y * rn.y, m.z * rn.z};
                pn[h] = V3<T>{z.x + beta * p.x, z.y + beta * p.y, z.z + beta * p.z};
                if (LM) { const V3<T> bb = ldv3(bV + o, i); accQ += (double)(T(0.5) * (d.x * (rn.x + bb.x))) + (double)(T(0.5) * (d.y * (rn.y + bb.y))) + (double)(T(0.5) * (d.z * (rn.z + bb.z))); }
                stv3(dOut + o, i, d);
            }
            stv3(rNew + o, i, rn); stv3(pNew + o, i, pn[h]);
        }
        P.D0[i] = Q4<T>{pn[0].x, pn[0].y, pn[0].z, pn[1].x}; P.D1[i] = Q4<T>{pn[1].y, pn[1].z, T(0), T(0)};
    }
    if (LM && L.q && !restart) {
        const double tq = blockReduceSum(accQ, scratch);
        if (threadIdx.x == 0) { if (L.qTag) storeTaggedPartial(L.q, blockIdx.x, tq, L.qTag); else L.q[blockIdx.x] = tq; }
    }
}

// (The persistent ARAP iteration of round 5 -- built, parity-green, break-even at 56.7 us -- is kept as a record under tools/round5/arap_onchip_experiment/ with its numbers in
// profiles/r05_arap_onchip_experiment.md; it is no longer wired into the library.  Round 6 measured what bounds the two kernels of an iteration with SQ / TCC / TCP counters and
// tried the record layout {p, M = sum_k p_a[k] dR/da_k, U} with id-only slots -- 25 % fewer VALU instructions and 21 % fewer HBM reads in the gather, same time; 38 MB more in the
// flat pass, 8 us slower: profiles/r06_arap_counters.md, tools/round6/arap_v2_pMU_records.patch -- before the plane layout above.)

// ---- the same for J^T F and diag(J^T J) (once per Gauss-Newton iteration) ---------------------------------------------------------
// Edge pass: rotation-derivative columns into the D planes (as arap_edges<2>) and one 9-scalar record per half-edge,
// {w res, w D_k . res, w^2 D_k . D_k}; vertex pass: adds the records of the vertex's out- and in-lists to what arap_vertices<2> wrote.
// With it the whole ARAP path is free of atomics: the same inputs give the same bits.
template <class T>
__global__ __launch_bounds__(kBlock) void arap_edgeJTF(ArapArgs<T> A, T* __restrict__ rec) {
    const long nE = A.nE;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < nE; e += (long)gridDim.x * blockDim.x) {
        const long a0 = A.v0[e], a1 = A.v1[e];
        const V3<T> O0 = ld3(A.Offset, a0), O1 = ld3(A.Offset, a1), ang = ld3(A.Angle, a0), U0 = ld3(A.UrShape, a0), U1 = ld3(A.UrShape, a1);
        const V3<T> u{U0.x - U1.x, U0.y - U1.y, U0.z - U1.z};
        V3<T> Ru{0, 0, 0}, D0, D1, D2;
        arap_rot(ang, u, Ru, D0, D1, D2);
        A.D[e] = D0.x; A.D[nE + e] = D0.y; A.D[2 * nE + e] = D0.z; A.D[3 * nE + e] = D1.x; A.D[4 * nE + e] = D1.y; A.D[5 * nE + e] = D1.z;
        A.D[6 * nE + e] = D2.x; A.D[7 * nE + e] = D2.y; A.D[8 * nE + e] = D2.z;
        const T w = A.w_reg, w2 = w * w;
        const V3<T> res{w * ((O0.x - O1.x) - Ru.x), w * ((O0.y - O1.y) - Ru.y), w * ((O0.z - O1.z) - Ru.z)};
        T* o = rec + 9 * e;
        o[0] = w * res.x; o[1] = w * res.y; o[2] = w * res.z;
        o[3] = w * dot3(D0, res); o[4] = w * dot3(D1, res); o[5] = w * dot3(D2, res);
        o[6] = w2 * dot3(D0, D0); o[7] = w2 * dot3(D1, D1); o[8] = w2 * dot3(D2, D2);
    }
}
template <class T>
__global__ __launch_bounds__(kBlock) void arap_vertexGatherJTF(ArapArgs<T> A, GraphCsr G, const T* __restrict__ rec, T* __restrict__ r, T* __restrict__ diag) {
    const long offA = 3 * A.N;
    const int slot = threadIdx.x % kLanesPerVertex;
    const long nGroups = (A.N + (kBlock / kLanesPerVertex) - 1) / (kBlock / kLanesPerVertex);
    for (long g = blockIdx.x; g < nGroups; g += gridDim.x) {
        const long i = g * (kBlock / kLanesPerVertex) + threadIdx.x / kLanesPerVertex;
        const bool ok = i < A.N;
        const long iv = ok ? i : 0;
        T s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int bo = G.outOff[iv], eo = ok ? G.outOff[iv + 1] : bo, bi = G.inOff[iv], ei = ok ? G.inOff[iv + 1] : bi;
        for (int k = 0; k < max(eo - bo, ei - bi); k += kLanesPerVertex) {
            const int ko = bo + slot + k, ki = bi + slot + k;
            const int eOut = ko < eo ? G.outIdx[ko] : -1, eIn = ki < ei ? G.inIdx[ki] : -1;
            const T* o = rec + 9 * (long)max(eOut, 0); const T* q = rec + 9 * (long)max(eIn, 0);
            const T mo = eOut >= 0 ? T(1) : T(0), mi = eIn >= 0 ? T(1) : T(0);
#pragma unroll
            for (int c = 0; c < 3; ++c) { s[c] += mi * q[c] - mo * o[c]; s[3 + c] += mo * o[3 + c]; s[6 + c] += mo * o[6 + c]; }   // r -= J^T F: head -w res, tail +w res
        }
#pragma unroll
        for (int m = 1; m < kLanesPerVertex; m <<= 1)
#pragma unroll
            for (int c = 0; c < 9; ++c) s[c] += __shfl_xor(s[c], m, kWave);
        if (ok && slot == 0) {
            const T w2 = A.w_reg * A.w_reg, deg = (T)((eo - bo) + (ei - bi));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                r[3 * i + c] += s[c]; r[offA + 3 * i + c] += s[3 + c];
                diag[3 * i + c] += w2 * deg; diag[offA + 3 * i + c] += s[6 + c];
            }
        }
    }
}

template <class T>
struct ArapOps : EnergyOps<T> {
    ArapArgs<T> A{};
    int cus = 256;
    long dCapacity = 0;
    // edge lists per vertex for the gather path (rebuilt when the graph arrays change)
    int *outOff = nullptr, *outIdx = nullptr, *inOff = nullptr, *inIdx = nullptr, *cursors = nullptr; T* Jp = nullptr;
    void* scanTemp = nullptr; size_t scanTempBytes = 0; unsigned long long* dChecksum = nullptr;
    const int *csrV0 = nullptr, *csrV1 = nullptr; int csrNE = -1; unsigned long long csrSum = 0; bool csrValid = false;
    bool useGather = true;   // (development builds: OPT_AMD_ARAP_GATHER=0 scatters with wave-aggregated atomics instead)
    T* D9 = nullptr; long d9Capacity = 0; int* nbr = nullptr;
    // symmetric-graph path (arap_applyEll): 16-byte planes D0, D1 (dynamic), T0, T1, U0 (per Gauss-Newton step) and the out-lists in ELL order; OPT_AMD_ARAP_SYM=0 keeps arap_applyFused
    bool useSym = true, symGraph = false;
    ArapPlanes<T> planes{}; void* planeMem = nullptr; int* ellMem = nullptr; int* dNotSym = nullptr;
    bool symPath() const { return useGather && useSym && symGraph; }
    ~ArapOps() override {
        if (D9) (void)hipFree(D9);
        if (nbr) (void)hipFree(nbr);
        for (void* q : {planeMem, (void*)ellMem, (void*)dNotSym}) if (q) (void)hipFree(q);
        for (void* q : {(void*)A.D, (void*)outOff, (void*)outIdx, (void*)inOff, (void*)inIdx, (void*)cursors, (void*)Jp, scanTemp, (void*)dChecksum}) if (q) (void)hipFree(q);
    }
    void ensureCsr(LaunchCtx& ctx) {
        hipStream_t st = ctx.stream;
        if (!dChecksum) HIP_CHECK(hipMalloc((void**)&dChecksum, 8));
        HIP_CHECK(hipMemsetAsync(dChecksum, 0, 8, st));
        const int ge = edgeGrid(A.nE, cus);
        csr_checksum<<<ge, kBlock, 0, st>>>(A.v0, A.v1, A.nE, dChecksum);
        unsigned long long sum = 0;
        HIP_CHECK(hipMemcpyAsync(&sum, dChecksum, 8, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipStreamSynchronize(st));
        if (csrValid && csrV0 == A.v0 && csrV1 == A.v1 && csrNE == A.nE && csrSum == sum) return;
        ScopedKernel k(ctx, "buildEdgeLists");
        for (void* q : {(void*)outOff, (void*)outIdx, (void*)inOff, (void*)inIdx, (void*)cursors, (void*)Jp}) if (q) HIP_CHECK(hipFree(q));
        const size_t nv = (size_t)A.N + 1;
        HIP_CHECK(hipMalloc((void**)&outOff, nv * 4)); HIP_CHECK(hipMalloc((void**)&inOff, nv * 4)); HIP_CHECK(hipMalloc((void**)&cursors, 2 * nv * 4));
        HIP_CHECK(hipMalloc((void**)&outIdx, (size_t)std::max(1, A.nE) * 4)); HIP_CHECK(hipMalloc((void**)&inIdx, (size_t)std::max(1, A.nE) * 4));
        HIP_CHECK(hipMalloc((void**)&Jp, (size_t)9 * std::max(1, A.nE) * sizeof(T)));       // 6-scalar records of J^T J p, 9-scalar records of J^T F
        HIP_CHECK(hipMemsetAsync(cursors, 0, 2 * nv * 4, st));
        int* outDeg = cursors; int* inDeg = cursors + nv;
        csr_count<<<ge, kBlock, 0, st>>>(A.v0, A.v1, A.nE, outDeg, inDeg);
        size_t need = 0;
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, outDeg, outOff, (int)nv, st));
        if (need > scanTempBytes) { if (scanTemp) HIP_CHECK(hipFree(scanTemp)); HIP_CHECK(hipMalloc(&scanTemp, need)); scanTempBytes = need; }
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTemp, need, outDeg, outOff, (int)nv, st));
        HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTemp, need, inDeg, inOff, (int)nv, st));
        HIP_CHECK(hipMemsetAsync(cursors, 0, 2 * nv * 4, st));
        csr_fill<<<ge, kBlock, 0, st>>>(A.v0, A.v1, A.nE, outOff, inOff, outDeg, inDeg, outIdx, inIdx);
        csr_sort<<<vgrid(), kBlock, 0, st>>>(A.N, outOff, outIdx);
        csr_sort<<<vgrid(), kBlock, 0, st>>>(A.N, inOff, inIdx);
        if (nbr) HIP_CHECK(hipFree(nbr));
        HIP_CHECK(hipMalloc((void**)&nbr, (size_t)2 * std::max(1, A.nE) * 4));
        csr_neighbours<<<ge, kBlock, 0, st>>>(A.v0, A.v1, A.nE, outIdx, inIdx, nbr, nbr + std::max(1, A.nE));
        // does every edge come with its reverse (per vertex: out-neighbours == in-neighbours as multisets)?  Then J^T J p walks the out-lists only (arap_applySym)
        symGraph = false;
        if (useSym) {
            if (!dNotSym) HIP_CHECK(hipMalloc((void**)&dNotSym, 4));
            HIP_CHECK(hipMemsetAsync(dNotSym, 0, 4, st));
            csr_symmetric<<<vgrid(), kBlock, 0, st>>>(A.N, outOff, nbr, inOff, nbr + std::max(1, A.nE), dNotSym);
            int bad = 1;
            HIP_CHECK(hipMemcpyAsync(&bad, dNotSym, 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipStreamSynchronize(st));
            symGraph = bad == 0;
            for (void* q : {planeMem, (void*)ellMem}) if (q) HIP_CHECK(hipFree(q));
            planeMem = nullptr; ellMem = nullptr; planes = ArapPlanes<T>{};
            if (symGraph) {      // the longest out-list decides the ELL width; a vertex beyond kEllMax sends the graph to the edge-list gather
                HIP_CHECK(hipMemsetAsync(dNotSym, 0, 4, st));
                csr_maxDegree<<<vgrid(), kBlock, 0, st>>>(A.N, outOff, dNotSym);
                int K = 0;
                HIP_CHECK(hipMemcpyAsync(&K, dNotSym, 4, hipMemcpyDeviceToHost, st)); HIP_CHECK(hipStreamSynchronize(st));
                if (K > kEllMax) symGraph = false;
                else {
                    K = std::max(K, 1);
                    const size_t n = (size_t)std::max<long>(1, A.N);
                    HIP_CHECK(hipMalloc(&planeMem, 5 * n * sizeof(Q4<T>))); HIP_CHECK(hipMemsetAsync(planeMem, 0, 5 * n * sizeof(Q4<T>), st));
                    HIP_CHECK(hipMalloc((void**)&ellMem, ((size_t)K + 1) * n * sizeof(int)));
                    Q4<T>* q = (Q4<T>*)planeMem;
                    planes = ArapPlanes<T>{q, q + n, q + 2 * n, q + 3 * n, q + 4 * n, ellMem, ellMem + (size_t)K * n, K};
                    arap_buildEll<<<vgrid(), kBlock, 0, st>>>(A.N, outOff, nbr, K, ellMem, ellMem + (size_t)K * n);
                }
            }
        }
        csrV0 = A.v0; csrV1 = A.v1; csrNE = A.nE; csrSum = sum; csrValid = true;
    }
    ArapOps(const unsigned* dims) {
        A.N = dims[0];
        this->usePreconditioner = true; this->usesGraph = true;                  // arap_mesh_deformation.t:9
        this->addUnknown(2, A.N, 3); this->addUnknown(3, A.N, 3);                // Offset, Angle (:4-5)
        int dev = 0; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        useGather = devSwitch("OPT_AMD_ARAP_GATHER", 1) != 0;
        fusedIterEnv = devSwitch("OPT_AMD_ARAP_ITER", -1);
        if (const char* e = getenv("OPT_AMD_ARAP_SYM")) useSym = atoi(e) != 0;
        if (const char* e = getenv("OPT_AMD_ARAP_SYM_XCD")) symXcd = atoi(e);
        if (const char* e = getenv("OPT_AMD_ARAP_VGRID")) symGridCap = atoi(e);
        if (const char* e = getenv("OPT_AMD_ARAP_WALK")) flatBackward = atoi(e);
    }
    bool bindInvariantDuringSolve() const override { return true; }      // the edge lists (and their checksum + read-back) depend on the graph arrays only
    void bind(void** p, LaunchCtx& ctx) override {
        A.w_fit = (T) * (const float*)p[0]; A.w_reg = (T) * (const float*)p[1];
        A.Offset = (const T*)p[2]; A.Angle = (const T*)p[3]; A.UrShape = (const T*)p[4]; A.Constraints = (const T*)p[5];
        A.nE = *(const int*)p[6]; A.v0 = (const int*)p[7]; A.v1 = (const int*)p[8];   // Graph("G", 6, "v0", {N}, 7, "v1", {N}, 8)
        if (A.nE > dCapacity) {
            if (A.D) HIP_CHECK(hipFree(A.D));
            dCapacity = A.nE;
            HIP_CHECK(hipMalloc((void**)&A.D, (size_t)9 * dCapacity * sizeof(T)));
            HIP_CHECK(hipMemset(A.D, 0, (size_t)9 * dCapacity * sizeof(T)));
        }
        if (useGather) ensureCsr(ctx);      // (UrShape is an input of the solve: arap_buildStatic reads it afresh in every Gauss-Newton step)
    }
    T* unknownPtr(int img) const override { return const_cast<T*>(img == 0 ? A.Offset : A.Angle); }
    int vgrid() const { static const int cap = getenv("OPT_AMD_ARAP_VGRID") ? atoi(getenv("OPT_AMD_ARAP_VGRID")) : kMaxPartials / 2; return (int)std::max<long>(1, std::min<long>((A.N + kBlock - 1) / kBlock, cap)); }
    void evalCost(Reduction& out, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(A.nE, cus);
        { ScopedKernel k(ctx, "computeCost"); arap_vertices<T, 0><<<gv, kBlock, 0, ctx.stream>>>(A, nullptr, nullptr, nullptr, nullptr, out.partials); }
        { ScopedKernel k(ctx, "computeCost_Graph"); arap_edges<T, 0><<<ge, kBlock, 0, ctx.stream>>>(A, nullptr, nullptr, nullptr, out.partials + gv); }
        out.n = gv + ge;
    }
    void evalJTF(T* r, T* diag, LaunchCtx& ctx) override {
        { ScopedKernel k(ctx, "PCGInit1"); arap_vertices<T, 2><<<vgrid(), kBlock, 0, ctx.stream>>>(A, nullptr, r, diag, nullptr, nullptr); }
        if (useGather) {
            GraphCsr G{outOff, outIdx, inOff, inIdx};
            { ScopedKernel k(ctx, "PCGInit1_Graph"); arap_edgeJTF<T><<<edgeGrid(A.nE, cus), kBlock, 0, ctx.stream>>>(A, Jp); }
            { ScopedKernel k(ctx, "PCGInit1_Gather"); arap_vertexGatherJTF<T><<<vgrid(), kBlock, 0, ctx.stream>>>(A, G, Jp, r, diag); }
            if (symPath()) {    // the sines / cosines of this Gauss-Newton iteration's angles and the rest positions as planes, for arap_applyEll
                ScopedKernel k(ctx, "vertexRecords");
                arap_buildStatic<T><<<vgrid(), kBlock, 0, ctx.stream>>>(A, planes);
            }
            if (!symPath()) {     // the derivative columns of this Gauss-Newton iteration as 36-byte rows for arap_applyFused
                if (A.nE > d9Capacity) { if (D9) HIP_CHECK(hipFree(D9)); d9Capacity = A.nE; HIP_CHECK(hipMalloc((void**)&D9, (size_t)9 * std::max<long>(1, d9Capacity) * sizeof(T))); }
                ScopedKernel k(ctx, "packDerivativeRows");
                arap_packD<T><<<edgeGrid(A.nE, cus), kBlock, 0, ctx.stream>>>(A.D, D9, (long)A.nE);
            }
        } else { ScopedKernel k(ctx, "PCGInit1_Graph"); arap_edges<T, 2><<<edgeGrid(A.nE, cus), kBlock, 0, ctx.stream>>>(A, nullptr, r, diag, nullptr); }
    }
    void applyJTJ(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(A.nE, cus);
        if (symPath()) {
            { ScopedKernel k(ctx, "packVertexRecords"); arap_packPlanes<T><<<gv, kBlock, 0, ctx.stream>>>(v, planes, A.N); }
            ScopedKernel k(ctx, "PCGStep1");
            launchSym(v, out, CtC, dot, ctx);
            return;
        }
        if (useGather) {
            GraphCsr G{outOff, outIdx, inOff, inIdx, nbr, nbr + std::max(1, A.nE)};
            ScopedKernel k(ctx, "PCGStep1");
            arap_applyFused<T><<<gv, kBlock, 0, ctx.stream>>>(A, G, D9, v, out, CtC, dot ? dot->partials : nullptr);
            if (dot) dot->n = gv;
            return;
        }
        {
            { ScopedKernel k(ctx, "PCGStep1"); arap_vertices<T, 3><<<gv, kBlock, 0, ctx.stream>>>(A, v, out, nullptr, CtC, dot ? dot->partials : nullptr); }
            { ScopedKernel k(ctx, "PCGStep1_Graph"); arap_edges<T, 3><<<ge, kBlock, 0, ctx.stream>>>(A, v, out, nullptr, dot ? dot->partials + gv : nullptr); }
        }
        if (dot) dot->n = gv + ge;
    }
    int symOcc = 0;
    int symGrid() {                     // one lane per vertex; all workgroups resident at once (what the gather's registers allow per CU: a second, partial round of workgroups costs
                                        // more than the longer walks -- 768 workgroups 26.5 us, 1952 31.5 us at 500 k vertices); a multiple of 8 for the XCD-aware order
        if (symOcc == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&symOcc, arap_applyEll<T, ARAP_ELL_BATCH>, kBlock, 0) != hipSuccess) { (void)hipGetLastError(); symOcc = 2; }
            symOcc = std::max(1, std::min(symOcc, 8));
        }
        const int cap = symGridCap;
        const long groups = (A.N + kBlock - 1) / kBlock;
        long g = std::max<long>(1, std::min<long>(groups, cap > 0 ? cap : std::min<long>((long)symOcc * cus, kMaxPartials / 2)));
        if (g >= 8) g -= g % 8;
        return (int)g;
    }
    int symGridCap = 0;                 // OPT_AMD_ARAP_VGRID (read per plan)
    int flatBackward = 1;               // OPT_AMD_ARAP_WALK=0: the flat pass walks its eighths forward like the gather (A/B switch, see XcdWalk)
    int symXcd = 1;                     // OPT_AMD_ARAP_SYM_XCD=0: consecutive vertex groups on consecutive workgroups (A/B switch, see arap_applyEll)
    int launchSym(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx, const ArapIterSums& S = ArapIterSums{nullptr, nullptr, nullptr, nullptr, nullptr}) {
        const int g = symGrid();
        double* part = dot ? dot->partials : nullptr;
        arap_applyEll<T, ARAP_ELL_BATCH><<<g, kBlock, 0, ctx.stream>>>(A, planes, v, out, CtC, part, symXcd && g % 8 == 0, S);
        if (dot) dot->n = g;
        return g;
    }
    // PCGStep3 of the previous iteration + PCGStep1 (symmetric-graph path): the flat pass that forms p = z + beta p also writes it into the vertex records the gather reads
    bool applyJTJFused(const T* pOld, const T* z, T* pNew, T* out, const T* CtC, Reduction* dot, const Reduction& bNum, const double* aNumOld, double* aNumNext, LaunchCtx& ctx) override {
        if (!symPath()) return false;
        { ScopedKernel k(ctx, "PCGStep3"); arap_step3Planes<T><<<vgrid(), kBlock, 0, ctx.stream>>>(z, pOld, pNew, planes, A.N, bNum.partials, bNum.n, aNumOld, aNumNext); }
        ScopedKernel k(ctx, "PCGStep1");
        launchSym(pNew, out, CtC, dot, ctx);
        return true;
    }
    // ---- two kernels per Gauss-Newton PCG iteration instead of three on the symmetric-graph path: [PCGStep2 + PCGStep3 of iteration k-1 as one flat pass that also rewrites the
    // planes: arap_flatStepPlanes] + [PCGStep1 of iteration k with the sums of the expanded beta numerator: arap_applyEll].  (Development builds: OPT_AMD_ARAP_ITER=0 keeps the reference's three
    // kernels per iteration.)  On the edge-list gather of asymmetric graphs the same fusion lost in three formulations (profiles/NOTES.md) and is not offered.
    int fusedIterEnv = -1;
    bool pcgIteration(const PcgIterArgs<T>& a, LaunchCtx& ctx) override {
        // Gauss-Newton and (round 6) Levenberg-Marquardt: a.CtC adds CtC p to the gather, the flat pass carries b, Q and the restart after a residual reset (ArapLmStep)
        const bool lmv = a.CtC != nullptr;
        if (symPath() && fusedIterEnv != 0 && a.pre && !this->slab.active && (!lmv || (a.b && a.q))) {
            const long n = 6 * A.N, nPad = (n + 3) / 4 * 4;
            const int g = symGrid();      // the gather's grid and vertex -> XCD mapping (XcdWalk)
            const int fx = symXcd && g % 8 == 0;
            if (a.first) {      // the solver adopts rNew / pNew after every launch: the start state moves there unchanged
                HIP_CHECK(hipMemcpyAsync(a.rNew, a.rOld, nPad * sizeof(T), hipMemcpyDeviceToDevice, ctx.stream));
                HIP_CHECK(hipMemcpyAsync(a.pNew, a.pOld, nPad * sizeof(T), hipMemcpyDeviceToDevice, ctx.stream));
                ScopedKernel k(ctx, "packVertexRecords"); arap_packPlanes<T><<<vgrid(), kBlock, 0, ctx.stream>>>(a.pNew, planes, A.N);
            } else if (lmv) {
                ScopedKernel k(ctx, "PCGStep2+PCGStep3");
                const ArapLmStep L{a.b, a.deltaOut ? a.deltaOut : a.delta, a.q->partials, a.qTag, a.afterReset, a.betaNum.partials, a.betaNum.n, a.betaDen.partials, a.betaDen.n};
                arap_flatStepPlanes<T, true><<<g, kBlock, 0, ctx.stream>>>(planes, A.N, a.delta, a.pOld, a.rOld, a.ApOld, a.pre, a.rNew, a.pNew, a.aNumPrev.partials, a.aNumPrev.n, a.aDenPrev.partials, a.aDenPrev.n,
                                                                       a.s2Prev.partials, a.s2Prev.n, a.s3Prev.partials, a.s3Prev.n, L, fx, flatBackward);
                if (!a.afterReset) a.q->n = g;
            } else {
                ScopedKernel k(ctx, "PCGStep2+PCGStep3");
                arap_flatStepPlanes<T, false><<<g, kBlock, 0, ctx.stream>>>(planes, A.N, a.delta, a.pOld, a.rOld, a.ApOld, a.pre, a.rNew, a.pNew, a.aNumPrev.partials, a.aNumPrev.n, a.aDenPrev.partials, a.aDenPrev.n,
                                                                        a.s2Prev.partials, a.s2Prev.n, a.s3Prev.partials, a.s3Prev.n, ArapLmStep{}, fx, flatBackward);
            }
            ScopedKernel k(ctx, "PCGStep1");
            const int gs = launchSym(a.pNew, a.ApNew, a.CtC, a.aDen, ctx, ArapIterSums{a.rNew, a.pre, a.aNum->partials, a.s2->partials, a.s3->partials});
            a.aNum->n = a.aDen->n = a.s2->n = a.s3->n = gs;
            return true;
        }
        return false;
    }
    void evalModelCost(const T* delta, Reduction& out, LaunchCtx& ctx) override {
        const int gv = vgrid(), ge = edgeGrid(A.nE, cus);
        { ScopedKernel k(ctx, "computeModelCost"); arap_vertices<T, 1><<<gv, kBlock, 0, ctx.stream>>>(A, delta, nullptr, nullptr, nullptr, out.partials); }
        { ScopedKernel k(ctx, "computeModelCost_Graph"); arap_edges<T, 1><<<ge, kBlock, 0, ctx.stream>>>(A, delta, nullptr, nullptr, out.partials + gv); }
        out.n = gv + ge;
    }
};

// volumetric_mesh_deformation (examples/volumetric_mesh_deformation/volumetric_mesh_deformation.t:1-20) on ARAP's kernels: every in-bounds lattice neighbour n of voxel c, in the
// .t's stencil order (+x, -x, +y, -y, +z, -z), is the half-edge (c -> n) -- Select(InBounds(0,0,0), Select(InBounds(n), edge, 0), 0) keeps exactly the edges between existing voxels --
// and the parameter slots are those of the volumetric .t re-ordered into ARAP's.  The lattice graph is symmetric (six neighbours), so J^T J p runs on the plane gather (arap_applyEll).
template <class T>
struct VolumetricArapOps : ArapOps<T> {
    int* dv0 = nullptr; int* dv1 = nullptr; int nEdges = 0;
    static const unsigned* count(const unsigned* dims) { static thread_local unsigned n[1]; n[0] = dims[0] * dims[1] * dims[2]; return n; }
    explicit VolumetricArapOps(const unsigned* dims) : ArapOps<T>(count(dims)) {
        const int W = (int)dims[0], H = (int)dims[1], D = (int)dims[2];
        std::vector<int> v0, v1;
        v0.reserve((size_t)6 * W * H * D); v1.reserve((size_t)6 * W * H * D);
        static const int off[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
        for (int z = 0; z < D; ++z) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
            const int c = (z * H + y) * W + x;
            for (const auto& o : off) {
                const int xx = x + o[0], yy = y + o[1], zz = z + o[2];
                if (xx < 0 || xx >= W || yy < 0 || yy >= H || zz < 0 || zz >= D) continue;
                v0.push_back(c); v1.push_back((zz * H + yy) * W + xx);
            }
        }
        nEdges = (int)v0.size();
        HIP_CHECK(hipMalloc((void**)&dv0, std::max<size_t>(1, v0.size()) * sizeof(int))); HIP_CHECK(hipMalloc((void**)&dv1, std::max<size_t>(1, v1.size()) * sizeof(int)));
        HIP_CHECK(hipMemcpy(dv0, v0.data(), v0.size() * sizeof(int), hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(dv1, v1.data(), v1.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    ~VolumetricArapOps() override { (void)hipFree(dv0); (void)hipFree(dv1); }
    void bind(void** p, LaunchCtx& ctx) override {      // volumetric: Offset, Angle, UrShape, Constraints, w_fitSqrt, w_regSqrt  ->  ARAP: w_fit, w_reg, Offset, Angle, UrShape, Constraints, |E|, v0, v1
        void* q[9] = {p[4], p[5], p[0], p[1], p[2], p[3], &nEdges, dv0, dv1};
        ArapOps<T>::bind(q, ctx);
    }
};

template <class T> EnergyOps<T>* makeCF(const unsigned* dims) { return new CurveFittingOps<T>(dims); }
template <class T> EnergyOps<T>* makeArap(const unsigned* dims) { return new ArapOps<T>(dims); }

}  // namespace

template <class T> EnergyOps<T>* makeVolumetricOnArap(const unsigned* dims) { return new VolumetricArapOps<T>(dims); }
template EnergyOps<float>* makeVolumetricOnArap<float>(const unsigned*);
template EnergyOps<double>* makeVolumetricOnArap<double>(const unsigned*);

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
    e.name = "arap_mesh_deformation"; e.nDims = 1; e.usePreconditioner = true; e.floatOnly = false; e.residualsPerElement = 3; e.residualsPerEdge = 3;
    e.params = {{ParamDecl::kScalar, "w_fitSqrt", "float", 0}, {ParamDecl::kScalar, "w_regSqrt", "float", 1},
                {ParamDecl::kUnknown, "Offset", "opt_float3", 2}, {ParamDecl::kUnknown, "Angle", "opt_float3", 3},
                {ParamDecl::kArray, "UrShape", "opt_float3", 4}, {ParamDecl::kArray, "Constraints", "opt_float3", 5},
                {ParamDecl::kGraphCount, "G", "int", 6}, {ParamDecl::kGraphIndex, "G.v0", "int", 7}, {ParamDecl::kGraphIndex, "G.v1", "int", 8}};
    e.makeFloat = makeArap<float>; e.makeDouble = makeArap<double>;
    return e;
}

}  // namespace optamd
