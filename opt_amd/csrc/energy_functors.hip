// The stencil energies served by the functor engine (stencil_engine.h): optical_flow, intrinsic_image_decomposition,
// volumetric_mesh_deformation -- SURVEY.md 8(f) rank 3.  Each functor restates the residuals of its reference .t once,
// against a scalar type S that the engine instantiates as T or as a dual number.
#include "stencil_engine.h"
#include "stencil_march.h"
#include "stencil_onchip.h"
#include "graph_common.h"      // makeVolumetricOnArap

namespace optamd {
namespace {

// ------------------------------------------------------------------------------------------------------------------
// examples/optical_flow/optical_flow.t:1-19.  X float2 flow; e_fit = w_fit (I(0,0) - I_hat(i + u, j + v)) with I_hat a
// SampledImage (bilinear, o.t:578-589) whose partials are the sampled derivative images (o.t:2486-2501);
// e_reg = w_reg (X(0,0) - X(n)) for the 4-neighbourhood, Select(InBounds(n), ., 0).  UsePreconditioner(false).
template <class T>
struct OpticalFlowE {
    static constexpr int NDIM = 2, NIMG = 1, K = 2, R = 9, NOFF = 5, NAUX = 0;
    static constexpr __host__ __device__ int off(int i, int a) { return a == 0 ? (i == 1 ? 1 : i == 2 ? -1 : 0) : a == 1 ? (i == 3 ? 1 : i == 4 ? -1 : 0) : 0; }
    static constexpr __host__ __device__ int imgOf(int) { return 0; }
    static constexpr __host__ __device__ int chOf(int k) { return k; }
    static constexpr __host__ __device__ int channels(int) { return 2; }
    static constexpr __host__ __device__ bool depends(int ri, int oi) { return oi == 0 || (ri >= 1 && (ri - 1) / 2 + 1 == oi); }
    static int unknownParam(int) { return 2; }
    int W, H, D;
    const T* X[NIMG];
    const T *I, *Ihat, *Idx, *Idy;
    T w_fit, w_reg; T* aux;
    __device__ void computeAux(int, int, int, T*) const {}
    void bindParams(void** p) {
        w_fit = (T) * (const float*)p[0]; w_reg = (T) * (const float*)p[1];
        X[0] = (const T*)p[2]; I = (const T*)p[3]; Ihat = (const T*)p[4]; Idx = (const T*)p[5]; Idy = (const T*)p[6];
    }
    __device__ bool excluded(int, int, int) const { return false; }
    __device__ T get(const T* im, int x, int y) const { return (x >= 0 && x < W && y >= 0 && y < H) ? im[(long)y * W + x] : T(0); }   // Image:get (o.t:570-576)
    __device__ T sample(const T* im, T x, T y) const {                                                                            // Image:sample (o.t:582-589)
        const int x0 = (int)floor(x), x1 = (int)ceil(x), y0 = (int)floor(y), y1 = (int)ceil(y);
        const T xn = x - T(x0), yn = y - T(y0);
        const T u = (T(1) - xn) * get(im, x0, y0) + xn * get(im, x1, y0);
        const T b = (T(1) - xn) * get(im, x0, y1) + xn * get(im, x1, y1);
        return (T(1) - yn) * u + yn * b;
    }
    template <class S, class C>
    __device__ __forceinline__ void residuals(const C& Xc, int x, int y, int, S* r) const {
        const S u = Xc(0), v = Xc(1);
        const T px = T(x) + valueOf(u), py = T(y) + valueOf(v);
        const S ih = chain2(sample(Ihat, px, py), sample(Idx, px, py), sample(Idy, px, py), u, v);
        r[0] = w_fit * (I[(long)y * W + x] - ih);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int dx = off(n + 1, 0), dy = off(n + 1, 1);
            const bool inb = Xc.in(dx, dy, 0);
            const S ex = w_reg * (u - Xc(0, dx, dy)), ey = w_reg * (v - Xc(1, dx, dy));
            r[1 + 2 * n] = inb ? ex : S(T(0)); r[2 + 2 * n] = inb ? ey : S(T(0));
        }
    }
};

// ------------------------------------------------------------------------------------------------------------------
// examples/intrinsic_image_decomposition/intrinsic_image_decomposition.t:1-31.  Unknowns r (float3, log-albedo) and s (float,
// log-shading).  Albedo smoothness is an L_p norm through lib.t:106-114: sqrt((|r_c - r_n| + 1e-7)^(p-2)) is a ComputedArray --
// a constant of the linearisation, re-evaluated from the current r after every update (computeAux / the engine's precompute
// kernel: four planes, so the PCG loop evaluates no pow) -- times (r_c - r_n); shading smoothness is plain; fit r + s - i.
// No UsePreconditioner call -> false; no Exclude.
template <class T>
struct IntrinsicE {
    static constexpr int NDIM = 2, NIMG = 2, K = 4, R = 19, NOFF = 5, NAUX = 4;      // four L_p weight planes, one per stencil direction
    static constexpr __host__ __device__ int off(int i, int a) { return a == 0 ? (i == 1 ? 1 : i == 2 ? -1 : 0) : a == 1 ? (i == 3 ? 1 : i == 4 ? -1 : 0) : 0; }
    static constexpr __host__ __device__ int imgOf(int k) { return k < 3 ? 0 : 1; }
    static constexpr __host__ __device__ int chOf(int k) { return k < 3 ? k : 0; }
    static constexpr __host__ __device__ int channels(int img) { return img == 0 ? 3 : 1; }
    // rows 0..11: albedo (direction n = ri / 3), 12..15: shading (direction ri - 12), 16..18: fit (centre only)
    static constexpr __host__ __device__ bool depends(int ri, int oi) { return oi == 0 || (ri < 12 && ri / 3 + 1 == oi) || (ri >= 12 && ri < 16 && ri - 12 + 1 == oi); }
    static int unknownParam(int img) { return img == 0 ? 4 : 6; }
    int W, H, D;
    const T* X[NIMG];
    const T* target;
    T w_fit, w_regA, w_regS, pNorm; T* aux;
    // the ComputedArray of L_p (lib.t:106-114) for stencil direction n at pixel (x, y): sqrt((|r_c - r_n| + 1e-7)^(p - 2)), from the current r
    __device__ void computeAux(int x, int y, int, T* out) const {
        const ValueCtx<T, IntrinsicE> V(*this, x, y, 0);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int dx = off(n + 1, 0), dy = off(n + 1, 1);
            const T c0 = V(0) - V(0, dx, dy), c1 = V(1) - V(1, dx, dy), c2 = V(2) - V(2, dx, dy);
            const T dist = sqrt(c0 * c0 + c1 * c1 + c2 * c2);                                   // L_2_norm(val_const)
            out[n] = sqrt(pow(dist + T(0.0000001), pNorm - T(2)));                             // lib.t:108-110
        }
    }
    void bindParams(void** p) {
        w_fit = (T) * (const float*)p[0]; w_regA = (T) * (const float*)p[1]; w_regS = (T) * (const float*)p[2];
        pNorm = *(const T*)p[3];                      // Param("pNorm", opt_float, 3)
        X[0] = (const T*)p[4]; target = (const T*)p[5]; X[1] = (const T*)p[6];
    }
    __device__ bool excluded(int, int, int) const { return false; }
    template <class S, class C>
    __device__ __forceinline__ void residuals(const C& Xc, int x, int y, int, S* r) const {
        const long N = (long)W * H, cpx = (long)y * W + x;
        const S rc[3] = {Xc(0), Xc(1), Xc(2)};
        const S sc = Xc(3);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int dx = off(n + 1, 0), dy = off(n + 1, 1);
            const bool inb = Xc.in(dx, dy, 0);
            const T sqrtC = aux[(long)n * N + cpx];                                            // the ComputedArray, frozen since the last precompute
#pragma unroll
            for (int c = 0; c < 3; ++c) { const S e = w_regA * (sqrtC * (rc[c] - Xc(c, dx, dy))); r[3 * n + c] = inb ? e : S(T(0)); }
            const S es = w_regS * (sc - Xc(3, dx, dy));
            r[12 + n] = inb ? es : S(T(0));
        }
        const long i = (long)y * W + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) r[16 + c] = w_fit * (rc[c] + sc - target[3 * i + c]);
    }
};

// ------------------------------------------------------------------------------------------------------------------
// examples/volumetric_mesh_deformation/volumetric_mesh_deformation.t:1-20.  3-D lattice, unknowns Offset and Angle (float3 each);
// fit (Offset - Constraints) where Constraints.x >= -999999.9; ARAP regulariser over the 6-neighbourhood with Rotate3D
// (lib.t:77-91), Select(InBounds(0,0,0), Select(InBounds(n), ., 0), 0).  UsePreconditioner(true).
template <class T>
struct VolumetricE {
    static constexpr int NDIM = 3, NIMG = 2, K = 6, R = 21, NOFF = 7, NAUX = 6;      // sin and cos of the three angles: every residual centre a voxel's gather
                                                                                     // visits needs them (21 sincos per voxel per J^T J p otherwise)
    static constexpr __host__ __device__ int off(int i, int a) {
        return a == 0 ? (i == 1 ? 1 : i == 2 ? -1 : 0) : a == 1 ? (i == 3 ? 1 : i == 4 ? -1 : 0) : (i == 5 ? 1 : i == 6 ? -1 : 0);
    }
    static constexpr __host__ __device__ int imgOf(int k) { return k < 3 ? 0 : 1; }
    static constexpr __host__ __device__ int chOf(int k) { return k % 3; }
    static constexpr __host__ __device__ int channels(int) { return 3; }
    static constexpr __host__ __device__ bool depends(int ri, int oi) { return oi == 0 || (ri >= 3 && (ri - 3) / 3 + 1 == oi); }
    static int unknownParam(int img) { return img; }
    int W, H, D;
    const T* X[NIMG];
    const T *Ur, *Cons;
    T w_fit, w_reg; T* aux;
    __device__ void computeAux(int x, int y, int z, T* out) const {                   // planes 0..2 sin, 3..5 cos of Angle at the voxel
        const long i = ((long)z * H + y) * W + x;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const T a = X[1][3 * i + k]; out[k] = sin(a); out[3 + k] = cos(a); }
    }
    void bindParams(void** p) {
        X[0] = (const T*)p[0]; X[1] = (const T*)p[1]; Ur = (const T*)p[2]; Cons = (const T*)p[3];
        w_fit = (T) * (const float*)p[4]; w_reg = (T) * (const float*)p[5];
    }
    __device__ bool excluded(int, int, int) const { return false; }
    template <class S, class C>
    __device__ __forceinline__ void residuals(const C& Xc, int x, int y, int z, S* r) const {
        const long i = ((long)z * H + y) * W + x;
        const S o[3] = {Xc(0), Xc(1), Xc(2)};
        const bool valid = Cons[3 * i] >= T(-999999.9);
#pragma unroll
        for (int c = 0; c < 3; ++c) { const S e = w_fit * (o[c] - Cons[3 * i + c]); r[c] = valid ? e : S(T(0)); }
        // Rotate3D(Angle, v): lib.t:77-91
        const S al = Xc(3), be = Xc(4), ga = Xc(5);
        // sin / cos come from the aux planes (same values as sin(al) ... evaluated here; refreshed after every update), their partials by the chain rule
        const long NV = (long)W * H * D;
        const T sav = aux[i], sbv = aux[NV + i], sgv = aux[2 * NV + i], cav = aux[3 * NV + i], cbv = aux[4 * NV + i], cgv = aux[5 * NV + i];
        const S sa = chain1(sav, cav, al), sb = chain1(sbv, cbv, be), sg = chain1(sgv, cgv, ga);
        const S ca = chain1(cav, -sav, al), cb = chain1(cbv, -sbv, be), cg = chain1(cgv, -sgv, ga);
        const S m0 = cg * cb, m1 = -sg * ca + cg * sb * sa, m2 = sg * sa + cg * sb * ca;
        const S m3 = sg * cb, m4 = cg * ca + sg * sb * sa, m5 = -cg * sa + sg * sb * ca;
        const S m6 = -sb, m7 = cb * sa, m8 = cb * ca;
#pragma unroll
        for (int n = 0; n < 6; ++n) {
            const int dx = off(n + 1, 0), dy = off(n + 1, 1), dz = off(n + 1, 2);
            const bool inb = Xc.in(dx, dy, dz);
            const long j = inb ? Xc.at(dx, dy, dz) : i;
            const T u0 = Ur[3 * i] - (inb ? Ur[3 * j] : T(0)), u1 = Ur[3 * i + 1] - (inb ? Ur[3 * j + 1] : T(0)), u2 = Ur[3 * i + 2] - (inb ? Ur[3 * j + 2] : T(0));
            const S e0 = w_reg * ((o[0] - Xc(0, dx, dy, dz)) - (m0 * u0 + m1 * u1 + m2 * u2));
            const S e1 = w_reg * ((o[1] - Xc(1, dx, dy, dz)) - (m3 * u0 + m4 * u1 + m5 * u2));
            const S e2 = w_reg * ((o[2] - Xc(2, dx, dy, dz)) - (m6 * u0 + m7 * u1 + m8 * u2));
            r[3 + 3 * n] = inb ? e0 : S(T(0)); r[4 + 3 * n] = inb ? e1 : S(T(0)); r[5 + 3 * n] = inb ? e2 : S(T(0));
        }
    }
};

// ---- optical_flow's Gauss-Newton PCG loop on the marching template (stencil_march.h) ----------------------------------------------------------------------------
// J^T J of optical_flow.t is a 5-point stencil with one 2 x 2 block per pixel: the data residual w_fit (I - I_hat(x + X)) contributes w_fit^2 g (g . p) with
// g = (I_hat_dx, I_hat_dy) sampled where the flow points (the partials SampledImage hands to the chain rule, o.t:582-589), every in-bounds smoothness edge appears in
// both directions (2 w_reg^2 (p_c - p_n) per neighbour, like poisson).  g is formed once per Gauss-Newton step (flow_coef: the functor's own sample()) and streamed as the
// operator's per-pixel coefficient; cost, J^T F and the LM loop stay on the functor engine.
template <class T>
struct FlowMarchOp {
    static constexpr int C = 2, kCoef = 2; static constexpr bool kMasked = false, kSplit31 = false;
    // variants that would spill are not instantiated (tests/test_kernel_resources.py reads the compiler's resource remarks): the widest marching workgroup, and the
    // on-chip (rows, waves, LM) combinations whose loop state does not fit 256 VGPRs
    static constexpr int kMaxBlock = 768;
    template <int R, int WV, bool LM> static constexpr bool spills() { return LM && WV == 8 && (sizeof(T) == 4 ? R == 16 : R == 8); }
    using Vec = MVec<T, 2>;
    T w_fit, w_reg;
    __device__ __forceinline__ Vec apply(const Vec& pc, const Vec& pl, const Vec& pr, const Vec& pu, const Vec& pd, bool hasL, bool hasR, bool hasU, bool hasD, const MVec<T, 2>& g) const {
        const T jp = -(w_fit * g.v[0]) * pc.v[0] + -(w_fit * g.v[1]) * pc.v[1];      // (J p) of the data residual: dr/dX = -w_fit g
        Vec o;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            T a = -(w_fit * g.v[c]) * jp;
            if (hasR) { const T t = w_reg * (w_reg * (pc.v[c] - pr.v[c])); a += t + t; }
            if (hasL) { const T t = w_reg * (w_reg * (pc.v[c] - pl.v[c])); a += t + t; }
            if (hasD) { const T t = w_reg * (w_reg * (pc.v[c] - pd.v[c])); a += t + t; }
            if (hasU) { const T t = w_reg * (w_reg * (pc.v[c] - pu.v[c])); a += t + t; }
            o.v[c] = a;
        }
        return o;
    }
};
template <class T>
__global__ __launch_bounds__(kBlock) void flow_coef(OpticalFlowE<T> e, T* __restrict__ g) {
    const long n = (long)e.W * e.H;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % e.W), y = (int)(i / e.W);
        const T px = T(x) + e.X[0][2 * i], py = T(y) + e.X[0][2 * i + 1];
        g[2 * i] = e.sample(e.Idx, px, py); g[2 * i + 1] = e.sample(e.Idy, px, py);
    }
}
template <class T>
struct OpticalFlowOps : StencilOps<T, OpticalFlowE<T>> {
    MarchLoop<T> march; T* coef = nullptr; bool useMarch = true;
    OpticalFlowOps(const unsigned* dims) : StencilOps<T, OpticalFlowE<T>>(dims, false) { if (const char* e = getenv("OPT_AMD_FLOW_MARCH")) useMarch = atoi(e) != 0; if (useMarch) oc.template reserveFor<FlowMarchOp<T>>(this->e.W, this->e.H, this->cus); }
    ~OpticalFlowOps() override { if (coef) (void)hipFree(coef); }
    bool deltaMovable() const override { return useMarch && !this->slab.active; }      // (PcgSolver::deltaTrial)
    bool pcgIteration(const PcgIterArgs<T>& a, LaunchCtx& ctx) override {
        if (!useMarch || a.pre || a.CtC) return false;      // Gauss-Newton only
        const long n = (long)this->e.W * this->e.H;
        if (!coef) HIP_CHECK(hipMalloc((void**)&coef, (size_t)(2 * n) * sizeof(T)));
        if (a.first) { ScopedKernel k(ctx, "operatorCoefficients"); flow_coef<T><<<this->grid(), kBlock, 0, ctx.stream>>>(this->e, coef); }
        return march.launch(FlowMarchOp<T>{this->e.w_fit, this->e.w_reg}, this->e.W, this->e.H, nullptr, this->cus, a, ctx, coef);
    }
    const T* pcgFinish(const T*, T* delta, LaunchCtx& ctx) override { return march.finish(delta, 2L * this->e.W * this->e.H, this->cus, ctx); }
    // ---- the whole Gauss-Newton linear solve on chip (stencil_onchip.h); the operator's per-pixel coefficient is formed first, as for the marching loop ----
    OnchipMarch<T> oc;
    bool onChipWithoutPreconditioner() const override { return true; }
    bool pcgSolveOnChip(const T* r0, const T* p0, T* delta, int L, double* traceDev, const OnChipLm<T>* lm, LaunchCtx& ctx) override {
        if (!useMarch || traceDev || this->slab.active) return false;
        int sx, ty, G;
        if (!oc.enabled || oc.failed || (lm && (!lm->CtC || lm->resetPeriod < L)) ||
            !(lm ? oc.template select<FlowMarchOp<T>, true>(this->e.W, this->e.H, this->cus, sx, ty, G) : oc.template select<FlowMarchOp<T>, false>(this->e.W, this->e.H, this->cus, sx, ty, G))) return false;      // (before the coefficient pass is spent)
        const long n = (long)this->e.W * this->e.H;
        if (!coef) HIP_CHECK(hipMalloc((void**)&coef, (size_t)(2 * n) * sizeof(T)));
        { ScopedKernel k(ctx, "operatorCoefficients"); flow_coef<T><<<this->grid(), kBlock, 0, ctx.stream>>>(this->e, coef); }
        return oc.solve(FlowMarchOp<T>{this->e.w_fit, this->e.w_reg}, this->e.W, this->e.H, nullptr, coef, r0, p0, delta, const_cast<T*>(this->e.X[0]), L, this->cus, ctx, lm);
    }
    bool onChipFailed() override { return oc.failedNow(); }
    bool onChipFailedPeek() override { return oc.failedPeek(); }
    void onChipRearm(LaunchCtx& ctx) override { oc.rearm(ctx); }
    std::string describe(int L, bool lmv) override { return oc.template describe<FlowMarchOp<T>>(this->e.W, this->e.H, this->cus, useMarch ? L : 0, lmv, "march_pcgIter"); }
};
template <class T> EnergyOps<T>* makeFlow(const unsigned* dims) { return new OpticalFlowOps<T>(dims); }
// ---- intrinsic_image_decomposition's Gauss-Newton PCG loop on the marching template ------------------------------------------------------------------------------
// Per pixel four unknowns (log-albedo r, three channels, and log-shading s: two images in the solver's vectors, Op::kSplit31) and four operator coefficients: the
// L_p weights sqrtC of the four stencil directions (IntrinsicE::computeAux, refreshed by precompute after every update).  The residual of edge (c, n) appears centred
// at c and centred at n with the same weight bit for bit (|r_c - r_n| is symmetric), so an in-bounds edge contributes 2 (w_A sqrtC)^2 (p_c - p_n) per albedo channel and
// 2 w_S^2 (p_c - p_n) to the shading; the fit rows w_fit (r_k + s - i_k) couple the four unknowns of a pixel.
template <class T>
struct IntrinsicMarchOp {
    static constexpr int C = 4, kCoef = 4; static constexpr bool kMasked = false, kSplit31 = true;
    static constexpr int kMaxBlock = sizeof(T) == 8 ? 256 : 768;      // (double: 512 / 768 threads spill 76 / 428 B)
    template <int R, int WV, bool LM> static constexpr bool spills() { return sizeof(T) == 8 && !LM && R == 4 && WV == 8; }
    using Vec = MVec<T, 4>;
    T w_fit, w_regA, w_regS;
    __device__ __forceinline__ Vec apply(const Vec& pc, const Vec& pl, const Vec& pr, const Vec& pu, const Vec& pd, bool hasL, bool hasR, bool hasU, bool hasD, const MVec<T, 4>& w) const {
        Vec o{{0, 0, 0, 0}};
        auto edge = [&](bool has, const Vec& pn, T sqrtC) {      // stencil direction order of the .t: (+1,0), (-1,0), (0,+1), (0,-1)
            if (!has) return;
#pragma unroll
            for (int c = 0; c < 3; ++c) { const T e = w_regA * (sqrtC * (pc.v[c] - pn.v[c])); const T t = (w_regA * sqrtC) * e; o.v[c] += t + t; }
            const T es = w_regS * (pc.v[3] - pn.v[3]); const T ts = w_regS * es; o.v[3] += ts + ts;
        };
        edge(hasR, pr, w.v[0]); edge(hasL, pl, w.v[1]); edge(hasD, pd, w.v[2]); edge(hasU, pu, w.v[3]);
#pragma unroll
        for (int c = 0; c < 3; ++c) { const T f = w_fit * (pc.v[c] + pc.v[3]); const T t = w_fit * f; o.v[c] += t; o.v[3] += t; }
        return o;
    }
};
template <class T>
__global__ __launch_bounds__(kBlock) void intrinsic_coef(const T* __restrict__ aux, T* __restrict__ coef, long n) {      // four planes -> four values per pixel
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        coef[4 * i] = aux[i]; coef[4 * i + 1] = aux[n + i]; coef[4 * i + 2] = aux[2 * n + i]; coef[4 * i + 3] = aux[3 * n + i];
    }
}
template <class T>
struct IntrinsicOps : StencilOps<T, IntrinsicE<T>> {
    MarchLoop<T> march; T* coef = nullptr; bool useMarch = true;
    IntrinsicOps(const unsigned* dims) : StencilOps<T, IntrinsicE<T>>(dims, false) { if (const char* e = getenv("OPT_AMD_INTRINSIC_MARCH")) useMarch = atoi(e) != 0; if (useMarch) oc.template reserveFor<IntrinsicMarchOp<T>>(this->e.W, this->e.H, this->cus); }
    ~IntrinsicOps() override { if (coef) (void)hipFree(coef); }
    bool deltaMovable() const override { return useMarch && !this->slab.active; }      // (PcgSolver::deltaTrial)
    bool pcgIteration(const PcgIterArgs<T>& a, LaunchCtx& ctx) override {
        if (!useMarch || a.pre || a.CtC) return false;      // Gauss-Newton only
        const long n = (long)this->e.W * this->e.H;
        if (!coef) HIP_CHECK(hipMalloc((void**)&coef, (size_t)(4 * n) * sizeof(T)));
        if (a.first) { ScopedKernel k(ctx, "operatorCoefficients"); intrinsic_coef<T><<<this->grid(), kBlock, 0, ctx.stream>>>(this->e.aux, coef, n); }
        return march.launch(IntrinsicMarchOp<T>{this->e.w_fit, this->e.w_regA, this->e.w_regS}, this->e.W, this->e.H, nullptr, this->cus, a, ctx, coef);
    }
    const T* pcgFinish(const T*, T* delta, LaunchCtx& ctx) override { return march.finish(delta, 4L * this->e.W * this->e.H, this->cus, ctx); }
    // ---- the whole linear solve on chip (stencil_onchip.h); the operator's coefficients are repacked first, as for the marching loop ----
    OnchipMarch<T> oc;
    bool onChipWithoutPreconditioner() const override { return true; }
    bool pcgSolveOnChip(const T* r0, const T* p0, T* delta, int L, double* traceDev, const OnChipLm<T>* lm, LaunchCtx& ctx) override {
        if (!useMarch || traceDev || this->slab.active) return false;
        using Op = IntrinsicMarchOp<T>;
        int sx, ty, G;
        if (!oc.enabled || oc.failed || (lm && (!lm->CtC || lm->resetPeriod < L)) ||
            !(lm ? oc.template select<Op, true>(this->e.W, this->e.H, this->cus, sx, ty, G) : oc.template select<Op, false>(this->e.W, this->e.H, this->cus, sx, ty, G))) return false;
        const long n = (long)this->e.W * this->e.H;
        if (!coef) HIP_CHECK(hipMalloc((void**)&coef, (size_t)(4 * n) * sizeof(T)));
        { ScopedKernel k(ctx, "operatorCoefficients"); intrinsic_coef<T><<<this->grid(), kBlock, 0, ctx.stream>>>(this->e.aux, coef, n); }
        // (X += delta runs flat over the solver's layout: the two unknown images are consecutive there, but the CALLER's two arrays need not be -- the update stays with the solver)
        return oc.solve(Op{this->e.w_fit, this->e.w_regA, this->e.w_regS}, this->e.W, this->e.H, nullptr, coef, r0, p0, delta, (T*)nullptr, L, this->cus, ctx, lm);
    }
    bool onChipAppliedUpdate() const override { return false; }
    bool onChipGuardedUpdate(const T* delta, LaunchCtx& ctx) override {      // X += delta per unknown image, unless the launch raised its failure word
        if (!oc.bad || !oc.launched) return false;
        ScopedKernel k(ctx, "PCGLinearUpdate");
        for (size_t i = 0; i < this->unknowns.size(); ++i) {
            const auto& u = this->unknowns[i];
            const long cnt = u.elems * u.channels;
            const int grid = (int)std::max<long>(1, std::min<long>((cnt + kBlock - 1) / kBlock, 4096));
            march_applyDelta<T><<<grid, kBlock, 0, ctx.stream>>>(this->unknownPtr((int)i), delta + u.offset, cnt, oc.bad, oc.hostErr);
        }
        return true;
    }
    bool onChipFailed() override { return oc.failedNow(); }
    bool onChipFailedPeek() override { return oc.failedPeek(); }
    void onChipRearm(LaunchCtx& ctx) override { oc.rearm(ctx); }
    std::string describe(int L, bool lmv) override { return oc.template describe<IntrinsicMarchOp<T>>(this->e.W, this->e.H, this->cus, useMarch ? L : 0, lmv, "march_pcgIter"); }
};
template <class T> EnergyOps<T>* makeIntrinsic(const unsigned* dims) { return new IntrinsicOps<T>(dims); }
// OPT_AMD_VOLUMETRIC_ARAP=0: the functor engine; default: ARAP's kernel set on the lattice graph (graph_common.h makeVolumetricOnArap -- the same energy, 151 -> ... ms at 96^3)
template <class T> EnergyOps<T>* makeVolumetric(const unsigned* dims) {
    const char* e = getenv("OPT_AMD_VOLUMETRIC_ARAP");
    // ARAP's kernel set holds, per voxel, 6 half-edges with 9 derivative columns, a 64-byte record, slots and CSR arrays -- about 400 (float) / 800 (double) bytes
    // per voxel on the device plus two host index vectors of 6 |V| ints -- where the functor engine needs ~100: it is taken only while that fits an eighth of
    // the device's memory (and 6 |V| half-edges fit 32-bit indices); larger volumes stay on the functor engine instead of failing in hipMalloc.
    const unsigned long long nV = (unsigned long long)dims[0] * dims[1] * dims[2];
    size_t freeB = 0, totalB = 0;
    // (decided from the device's TOTAL memory: the same problem takes the same kernel set whatever else is allocated -- ADVICE round 4)
    const bool fits = hipMemGetInfo(&freeB, &totalB) == hipSuccess && nV * (sizeof(T) == 8 ? 800ull : 400ull) < totalB / 8;
    if ((!e || atoi(e) != 0) && nV < (1ull << 28) && fits) return makeVolumetricOnArap<T>(dims);
    return new StencilOps<T, VolumetricE<T>>(dims, true);
}

}  // namespace

EnergyInfo opticalFlowInfo() {
    EnergyInfo e;
    e.name = "optical_flow"; e.nDims = 2; e.usePreconditioner = false; e.floatOnly = false;
    e.params = {{ParamDecl::kScalar, "w_fit", "float", 0}, {ParamDecl::kScalar, "w_reg", "float", 1}, {ParamDecl::kUnknown, "X", "opt_float2", 2},
                {ParamDecl::kArray, "I", "opt_float", 3}, {ParamDecl::kArray, "I_hat", "opt_float", 4}, {ParamDecl::kArray, "I_hat_dx", "opt_float", 5},
                {ParamDecl::kArray, "I_hat_dy", "opt_float", 6}};
    e.makeFloat = makeFlow<float>; e.makeDouble = makeFlow<double>;
    return e;
}
EnergyInfo intrinsicInfo() {
    EnergyInfo e;
    e.name = "intrinsic_image_decomposition"; e.nDims = 2; e.usePreconditioner = false; e.floatOnly = false;
    e.params = {{ParamDecl::kScalar, "w_fitSqrt", "float", 0}, {ParamDecl::kScalar, "w_regSqrtAlbedo", "float", 1}, {ParamDecl::kScalar, "w_regSqrtShading", "float", 2},
                {ParamDecl::kScalar, "pNorm", "opt_float", 3}, {ParamDecl::kUnknown, "r", "opt_float3", 4}, {ParamDecl::kArray, "r_const", "opt_float3", 4},
                {ParamDecl::kArray, "i", "opt_float3", 5}, {ParamDecl::kUnknown, "s", "opt_float", 6}};
    e.makeFloat = makeIntrinsic<float>; e.makeDouble = makeIntrinsic<double>;
    return e;
}
EnergyInfo volumetricInfo() {
    EnergyInfo e;
    e.name = "volumetric_mesh_deformation"; e.nDims = 3; e.usePreconditioner = true; e.floatOnly = false;
    e.params = {{ParamDecl::kUnknown, "Offset", "opt_float3", 0}, {ParamDecl::kUnknown, "Angle", "opt_float3", 1}, {ParamDecl::kArray, "UrShape", "opt_float3", 2},
                {ParamDecl::kArray, "Constraints", "opt_float3", 3}, {ParamDecl::kScalar, "w_fitSqrt", "float", 4}, {ParamDecl::kScalar, "w_regSqrt", "float", 5}};
    e.makeFloat = makeVolumetric<float>; e.makeDouble = makeVolumetric<double>;
    return e;
}

}  // namespace optamd
