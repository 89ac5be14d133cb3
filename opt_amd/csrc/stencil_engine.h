// Functor-driven kernel set for stencil energies on regular grids (2-D or 3-D index spaces).
//
// The hand-tuned kernel sets (image_warping, poisson, SFS) are written per energy.  The remaining image examples of the
// reference share one shape -- a few residuals per pixel, each a short expression of the unknowns at the pixel and at a
// fixed set of stencil offsets (examples/optical_flow, intrinsic_image_decomposition, volumetric_mesh_deformation) -- and
// are served by this engine instead: the energy is a small device functor that writes its residuals ONCE against a
// scalar type S, and the engine instantiates it with
//     S = T                      cost                                   (o.t:2375-2385)
//     S = Dual<T, K>             evalJTF: gradient and diag(J^T J)      (o.t:2129-2172)
//     S = Dual<T, K + 1>         applyJTJ: slot 0 carries J v, slots 1..K the partials w.r.t. this pixel's unknowns (o.t:2029-2089)
//     S = Dual<T, 1>             model cost 1/2 (F + J delta)^2         (o.t:2174-2225)
// which is what Opt's generator derives symbolically from the .t (ad.t).  Unknown-centric gather like the generated code:
// thread c owns pixel c and visits the residuals centred at c - s for every stencil offset s, seeding the dual parts on the
// unknown at offset s of that centre -- the seed positions are compile-time constants, so after inlining the derivative
// slots of untouched unknowns fold away.  No atomics, deterministic, one thread per pixel; inputs are read through L1/L2.
// Semantics shared with the other kernel sets (energy.h): out-of-image loads return 0 (o.t:570-576); residuals centred on an
// excluded pixel count in J^T F / J^T J of their non-excluded neighbours but not in the cost (solver.t:583 vs o.t:2029-2089);
// rows of excluded unknowns are 0.  Whether a residual whose stencil leaves the image is dropped is the functor's business
// (it must restate the InBounds / default-zero rule of its .t).
#pragma once
#include "energy.h"

namespace optamd {

// ---- forward-mode dual numbers (device) --------------------------------------------------------------------------
template <class T, int N>
struct Dual {
    T v; T d[N];
    __device__ __forceinline__ Dual() : v(0) { for (int i = 0; i < N; ++i) d[i] = 0; }
    __device__ __forceinline__ Dual(T c) : v(c) { for (int i = 0; i < N; ++i) d[i] = 0; }
};
#define OPTAMD_DUAL_LOOP for (int i = 0; i < N; ++i)
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator+(const Dual<T, N>& a, const Dual<T, N>& b) { Dual<T, N> r; r.v = a.v + b.v; OPTAMD_DUAL_LOOP r.d[i] = a.d[i] + b.d[i]; return r; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator-(const Dual<T, N>& a, const Dual<T, N>& b) { Dual<T, N> r; r.v = a.v - b.v; OPTAMD_DUAL_LOOP r.d[i] = a.d[i] - b.d[i]; return r; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator-(const Dual<T, N>& a) { Dual<T, N> r; r.v = -a.v; OPTAMD_DUAL_LOOP r.d[i] = -a.d[i]; return r; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator*(const Dual<T, N>& a, const Dual<T, N>& b) { Dual<T, N> r; r.v = a.v * b.v; OPTAMD_DUAL_LOOP r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator+(const Dual<T, N>& a, T b) { Dual<T, N> r = a; r.v += b; return r; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator+(T a, const Dual<T, N>& b) { return b + a; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator-(const Dual<T, N>& a, T b) { Dual<T, N> r = a; r.v -= b; return r; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator-(T a, const Dual<T, N>& b) { return (-b) + a; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator*(const Dual<T, N>& a, T b) { Dual<T, N> r; r.v = a.v * b; OPTAMD_DUAL_LOOP r.d[i] = a.d[i] * b; return r; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator*(T a, const Dual<T, N>& b) { return b * a; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> sin(const Dual<T, N>& a) { Dual<T, N> r; T s, c; sincosT(a.v, &s, &c); r.v = s; OPTAMD_DUAL_LOOP r.d[i] = c * a.d[i]; return r; }   // ad.t:795
template <class T, int N> __device__ __forceinline__ Dual<T, N> cos(const Dual<T, N>& a) { Dual<T, N> r; T s, c; sincosT(a.v, &s, &c); r.v = c; OPTAMD_DUAL_LOOP r.d[i] = -s * a.d[i]; return r; }  // ad.t:787
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator/(const Dual<T, N>& a, const Dual<T, N>& b) { Dual<T, N> r; const T inv = T(1) / b.v; r.v = a.v * inv; OPTAMD_DUAL_LOOP r.d[i] = (a.d[i] - r.v * b.d[i]) * inv; return r; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator/(const Dual<T, N>& a, T b) { return a * (T(1) / b); }
template <class T, int N> __device__ __forceinline__ Dual<T, N> operator/(T a, const Dual<T, N>& b) { return Dual<T, N>(a) / b; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> sqrt(const Dual<T, N>& a) { Dual<T, N> r; r.v = ::sqrt(a.v); const T k = T(1) / (T(2) * r.v); OPTAMD_DUAL_LOOP r.d[i] = k * a.d[i]; return r; }   // ad.t:797
#undef OPTAMD_DUAL_LOOP
// scalar overloads next to the dual ones (a functor calls sin(x) / cos(x) on S = T as well; the templates above hide ::sin)
__device__ __forceinline__ float sin(float x) { return ::sinf(x); }
__device__ __forceinline__ double sin(double x) { return ::sin(x); }
__device__ __forceinline__ float cos(float x) { return ::cosf(x); }
__device__ __forceinline__ double cos(double x) { return ::cos(x); }
__device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
__device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
template <class T> __device__ __forceinline__ T valueOf(T x) { return x; }
template <class T, int N> __device__ __forceinline__ T valueOf(const Dual<T, N>& x) { return x.v; }
// f(u, v) with known partials (the SampledImage operator, o.t:2486-2501): value and chain rule
template <class T> __device__ __forceinline__ T chain2(T f, T, T, T, T) { return f; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> chain2(T f, T fu, T fv, const Dual<T, N>& u, const Dual<T, N>& v) {
    Dual<T, N> r; r.v = f; for (int i = 0; i < N; ++i) r.d[i] = fu * u.d[i] + fv * v.d[i]; return r;
}
// f(u) with known value and derivative (e.g. a sine tabulated in an aux plane)
template <class T> __device__ __forceinline__ T chain1(T f, T, T) { return f; }
template <class T, int N> __device__ __forceinline__ Dual<T, N> chain1(T f, T fu, const Dual<T, N>& u) {
    Dual<T, N> r; r.v = f; for (int i = 0; i < N; ++i) r.d[i] = fu * u.d[i]; return r;
}
template <class T> __device__ __forceinline__ T part0(T) { return T(0); }      // chain2 needs the arguments as S; scalars carry no partials

// ---- accessors handed to the functor ----------------------------------------------------------------------------------
// E (the energy functor type) provides, all constexpr / static:
//   K, R, NOFF, NIMG;  off(i, axis) the stencil offsets (i = 0 is the centre);  imgOf(k), chOf(k), channels(img);
//   depends(ri, oi): may residual ri depend on the unknowns at offset oi (lets the compiler drop the others);
// and members W, H, D, X[NIMG] (caller arrays of the unknown images), plus whatever inputs it needs.
template <class T, class E>
struct GridView {
    const E& e; int x, y, z;         // the centre the residuals are evaluated at
    __device__ __forceinline__ bool in(int dx, int dy, int dz) const {
        const int xx = x + dx, yy = y + dy, zz = z + dz;
        return xx >= 0 && xx < e.W && yy >= 0 && yy < e.H && zz >= 0 && zz < e.D;
    }
    __device__ __forceinline__ long at(int dx, int dy, int dz) const { return ((long)(z + dz) * e.H + (y + dy)) * e.W + (x + dx); }
};
// S = T: current values of the unknowns
template <class T, class E>
struct ValueCtx : GridView<T, E> {
    __device__ __forceinline__ ValueCtx(const E& e_, int x_, int y_, int z_) : GridView<T, E>{e_, x_, y_, z_} {}
    __device__ __forceinline__ T operator()(int k, int dx = 0, int dy = 0, int dz = 0) const {
        if (!this->in(dx, dy, dz)) return T(0);
        return this->e.X[E::imgOf(k)][this->at(dx, dy, dz) * E::channels(E::imgOf(k)) + E::chOf(k)];
    }
};
// S = Dual<T, N>.  DIR: slot 0 = directional derivative along `v` (a solver vector).  SEED >= 0: slots DIR..DIR+K-1 are the
// partials w.r.t. the K unknowns of the pixel at stencil offset SEED of this centre.
template <class T, class E, bool DIR, int SEED>
struct DualCtx : GridView<T, E> {
    static constexpr int N = (DIR ? 1 : 0) + (SEED >= 0 ? E::K : 0);
    const T* v; const long* voff;
    __device__ __forceinline__ DualCtx(const E& e_, int x_, int y_, int z_, const T* v_, const long* voff_) : GridView<T, E>{e_, x_, y_, z_}, v(v_), voff(voff_) {}
    __device__ __forceinline__ Dual<T, N> operator()(int k, int dx = 0, int dy = 0, int dz = 0) const {
        Dual<T, N> r;
        const bool inb = this->in(dx, dy, dz);
        const int img = E::imgOf(k);
        const long i = inb ? this->at(dx, dy, dz) * E::channels(img) + E::chOf(k) : 0;
        r.v = inb ? this->e.X[img][i] : T(0);
        if (DIR) r.d[0] = inb ? v[voff[img] + i] : T(0);
        if (SEED >= 0) { if (dx == E::off(SEED < 0 ? 0 : SEED, 0) && dy == E::off(SEED < 0 ? 0 : SEED, 1) && dz == E::off(SEED < 0 ? 0 : SEED, 2)) r.d[(DIR ? 1 : 0) + k] = T(1); }
        return r;
    }
};

template <class E> struct VOff { long o[E::NIMG]; };

// ComputedArrays (o.t:2387-2409): a functor with NAUX > 0 gets NAUX planes `aux` (plane k at aux + k * N) that the engine fills with
// computeAux() in EnergyOps::precompute -- after bind and after every update / revert, like the reference's precompute kernel
// (solver.t:607-614, 1005, 1116, 1155) -- and that residuals() may read as constants of the linearisation.
template <class T, class E>
__global__ __launch_bounds__(kBlock) void se_precompute(E e) {
    const long N = (long)e.W * e.H * e.D;
    for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < N; c += (long)gridDim.x * blockDim.x) {
        const int x = (int)(c % e.W), y = (int)((c / e.W) % e.H), z = (int)(c / ((long)e.W * e.H));
        T vals[E::NAUX > 0 ? E::NAUX : 1];
        e.computeAux(x, y, z, vals);
#pragma unroll
        for (int k = 0; k < E::NAUX; ++k) e.aux[(long)k * N + c] = vals[k];
    }
}

// ---- kernels --------------------------------------------------------------------------------------------------------
template <class T, class E>
__global__ __launch_bounds__(kBlock) void se_cost(E e, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    const long N = (long)e.W * e.H * e.D;
    double acc = 0;
    for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < N; c += (long)gridDim.x * blockDim.x) {
        const int x = (int)(c % e.W), y = (int)((c / e.W) % e.H), z = (int)(c / ((long)e.W * e.H));
        if (e.excluded(x, y, z)) continue;
        T r[E::R];
        e.template residuals<T>(ValueCtx<T, E>(e, x, y, z), x, y, z, r);
        T s = 0;
#pragma unroll
        for (int i = 0; i < E::R; ++i) s += r[i] * r[i];
        acc += (double)(T(0.5) * s);
    }
    const double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

template <class T, class E>
__global__ __launch_bounds__(kBlock) void se_modelCost(E e, const T* __restrict__ delta, VOff<E> vo, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    const long N = (long)e.W * e.H * e.D;
    double acc = 0;
    for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < N; c += (long)gridDim.x * blockDim.x) {
        const int x = (int)(c % e.W), y = (int)((c / e.W) % e.H), z = (int)(c / ((long)e.W * e.H));
        if (e.excluded(x, y, z)) continue;
        typedef Dual<T, 1> S;
        S r[E::R];
        e.template residuals<S>(DualCtx<T, E, true, -1>(e, x, y, z, delta, vo.o), x, y, z, r);
        T s = 0;
#pragma unroll
        for (int i = 0; i < E::R; ++i) { const T m = r[i].v + r[i].d[0]; s += m * m; }
        acc += (double)(T(0.5) * s);
    }
    const double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// residuals centred at c - off(OI), differentiated w.r.t. the unknowns of pixel c
template <class T, class E, int OI, bool JTJ>
struct GatherStep {
    static __device__ __forceinline__ void run(const E& e, int x, int y, int z, const T* v, const long* voff, T* g, T* d) {
        const int cx = x - E::off(OI, 0), cy = y - E::off(OI, 1), cz = z - E::off(OI, 2);
        if (cx >= 0 && cx < e.W && cy >= 0 && cy < e.H && cz >= 0 && cz < e.D) {
            typedef DualCtx<T, E, JTJ, OI> Ctx;
            typedef Dual<T, Ctx::N> S;
            S r[E::R];
            e.template residuals<S>(Ctx(e, cx, cy, cz, v, voff), cx, cy, cz, r);
#pragma unroll
            for (int i = 0; i < E::R; ++i) {
                if (!E::depends(i, OI)) continue;
#pragma unroll
                for (int k = 0; k < E::K; ++k) {
                    if (JTJ) g[k] += r[i].d[1 + k] * r[i].d[0];                  // (dr/dx_k) (J v)_r
                    else { g[k] += r[i].d[k] * r[i].v; d[k] += r[i].d[k] * r[i].d[k]; }
                }
            }
        }
        if constexpr (OI + 1 < E::NOFF) GatherStep<T, E, OI + 1, JTJ>::run(e, x, y, z, v, voff, g, d);
    }
};

// JTF: out = -J^T F, diag = diag(J^T J).   JTJ: out = J^T J v (+ CtC .* v), partials of v . out
template <class T, class E, bool JTJ>
__global__ __launch_bounds__(kBlock) void se_gather(E e, const T* __restrict__ v, VOff<E> vo, T* __restrict__ out, T* __restrict__ diag,
                                                    const T* __restrict__ CtC, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    const long N = (long)e.W * e.H * e.D;
    double acc = 0;
    for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < N; c += (long)gridDim.x * blockDim.x) {
        const int x = (int)(c % e.W), y = (int)((c / e.W) % e.H), z = (int)(c / ((long)e.W * e.H));
        T g[E::K], d[E::K];
#pragma unroll
        for (int k = 0; k < E::K; ++k) { g[k] = 0; d[k] = 0; }
        const bool ex = e.excluded(x, y, z);
        if (!ex) GatherStep<T, E, 0, JTJ>::run(e, x, y, z, v, vo.o, g, d);
#pragma unroll
        for (int k = 0; k < E::K; ++k) {
            const long i = vo.o[E::imgOf(k)] + c * E::channels(E::imgOf(k)) + E::chOf(k);
            if (JTJ) {
                T s = g[k];
                if (CtC) s += CtC[i] * v[i];
                if (ex) s = 0;
                out[i] = s;
                acc += (double)(v[i] * s);
            } else { out[i] = -g[k]; diag[i] = d[k]; }
        }
    }
    if (JTJ) {
        const double t = blockReduceSum(acc, scratch);
        if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
    }
}

// ---- EnergyOps on top of a functor ---------------------------------------------------------------------------------------
// E must also provide (host):  void bindParams(void** params)  and  static int unknownParam(int img).
template <class T, class E>
struct StencilOps : EnergyOps<T> {
    E e{};
    VOff<E> vo{};
    int cus = 256;
    explicit StencilOps(const unsigned* dims, bool usePre) {
        e.W = (int)dims[0]; e.H = E::NDIM >= 2 ? (int)dims[1] : 1; e.D = E::NDIM >= 3 ? (int)dims[2] : 1;
        this->usePreconditioner = usePre;
        const long n = (long)e.W * e.H * e.D;
        for (int i = 0; i < E::NIMG; ++i) { vo.o[i] = this->nScalars; this->addUnknown(E::unknownParam(i), n, E::channels(i)); }
        int dev = 0; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if constexpr (E::NAUX > 0) { HIP_CHECK(hipMalloc((void**)&e.aux, (size_t)E::NAUX * n * sizeof(T))); HIP_CHECK(hipMemset(e.aux, 0, (size_t)E::NAUX * n * sizeof(T))); }
    }
    ~StencilOps() override { if constexpr (E::NAUX > 0) (void)hipFree(e.aux); }
    void precompute(LaunchCtx& ctx) override {
        if constexpr (E::NAUX > 0) { ScopedKernel k(ctx, "precompute"); se_precompute<T, E><<<grid(), kBlock, 0, ctx.stream>>>(e); }
    }
    int grid() const { const long n = (long)e.W * e.H * e.D; return (int)std::max<long>(1, std::min<long>((n + kBlock - 1) / kBlock, std::min<long>(kMaxPartials, (long)cus * 8))); }
    void bind(void** p, LaunchCtx&) override { e.bindParams(p); }
    T* unknownPtr(int img) const override { return const_cast<T*>(e.X[img]); }
    void evalCost(Reduction& out, LaunchCtx& ctx) override { ScopedKernel k(ctx, "computeCost"); se_cost<T, E><<<grid(), kBlock, 0, ctx.stream>>>(e, out.partials); out.n = grid(); }
    void evalJTF(T* r, T* diag, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "PCGInit1");
        se_gather<T, E, false><<<grid(), kBlock, 0, ctx.stream>>>(e, nullptr, vo, r, diag, nullptr, nullptr);
    }
    void applyJTJ(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "PCGStep1");
        se_gather<T, E, true><<<grid(), kBlock, 0, ctx.stream>>>(e, v, vo, out, nullptr, CtC, dot ? dot->partials : nullptr);
        if (dot) dot->n = grid();
    }
    void evalModelCost(const T* delta, Reduction& out, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "computeModelCost"); se_modelCost<T, E><<<grid(), kBlock, 0, ctx.stream>>>(e, delta, vo, out.partials); out.n = grid();
    }
};

}  // namespace optamd
