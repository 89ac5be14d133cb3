// poisson_image_editing (gradient-domain blend) and laplacian (the reference's tests/minimal energy):
// linear 4-neighbour / 2-neighbour stencils over one unknown image.
//
// Energies restated (reference examples/poisson_image_editing/poisson_image_editing.t:1-13 and
// tests/minimal/laplacian.t:1-7):
//   poisson  : X float4 unknown, T float4, M mask.  r[c,n,k] = InBounds(c+n) * [(X_c - X_n) - (T_c - T_n)]_k for the 4
//              neighbours; UsePreconditioner(false); Exclude(M != 0).  Every in-bounds edge appears twice in J^T J
//              (the residual centred at c and the one centred at the neighbour, the latter also when the
//              neighbour is excluded -- SURVEY.md 8a "exclude" row), so
//                 J^T F (c) = 2 sum_n [(X_c - X_n) - (T_c - T_n)],  diag = 2 #n,  (J^T J p)(c) = 2 sum_n (p_c - p_n).
//   laplacian: X float unknown, A float.  r = { 0.2 (X - A), X(0,0) - X(1,0), X(0,0) - X(0,1) }; a residual that
//              leaves the image is 0 (o.t:1930-1933); no Exclude, no preconditioner.
// Both are HBM-bound streaming stencils (poisson: 36 B/pixel algorithmic in applyJTJ).  The once-per-step kernels are
// one-thread-per-pixel with direct neighbour loads (rows are W*16 B, so the +-1 row re-reads come from L2); the Gauss-Newton
// PCG loop runs on the marching template of stencil_march.h (one launch per iteration, 65 B/pixel in float).
#include "energy.h"
#include "stencil_march.h"
#include "stencil_onchip.h"

namespace optamd {
namespace {

struct F4 { float x, y, z, w; };
template <class T> struct V4 { T x, y, z, w; };
template <class T> __device__ __forceinline__ V4<T> operator-(const V4<T>& a, const V4<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
template <class T> __device__ __forceinline__ V4<T> operator+(const V4<T>& a, const V4<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
template <class T> __device__ __forceinline__ V4<T> operator*(T s, const V4<T>& a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
template <class T> __device__ __forceinline__ T dot4(const V4<T>& a, const V4<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

template <class T>
struct PArgs {
    int W, H;
    const T* X; const T* Tg; const T* M;
};

__device__ __forceinline__ int flatGridLoopBegin() { return blockIdx.x * blockDim.x + threadIdx.x; }

// mode 0: cost partials ; mode 1: model cost partials (needs delta)
template <class T, int MODE>
__global__ __launch_bounds__(kBlock) void poisson_cost(PArgs<T> A, const T* __restrict__ delta, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    const long N = (long)A.W * A.H;
    const V4<T>* X = (const V4<T>*)A.X; const V4<T>* Tg = (const V4<T>*)A.Tg; const V4<T>* D = (const V4<T>*)delta;
    double acc = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        if (A.M[i] != T(0)) continue;   // residuals centred on excluded pixels are not part of the cost (solver.t:583)
        const int x = (int)(i % A.W), y = (int)(i / A.W);
        const V4<T> xc = X[i], tc = Tg[i];
        V4<T> dc{0, 0, 0, 0}; if (MODE == 1) dc = D[i];
        T e = 0;
        const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nx = x + dx[n], ny = y + dy[n];
            if (nx < 0 || nx >= A.W || ny < 0 || ny >= A.H) continue;
            const long ni = (long)ny * A.W + nx;
            V4<T> r = (xc - X[ni]) - (tc - Tg[ni]);
            if (MODE == 1) r = r + (dc - D[ni]);
            e += dot4(r, r);
        }
        acc += (double)(T(0.5) * e);
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

template <class T>
__global__ __launch_bounds__(kBlock) void poisson_evalJTF(PArgs<T> A, T* __restrict__ r, T* __restrict__ diag) {
    const long N = (long)A.W * A.H;
    const V4<T>* X = (const V4<T>*)A.X; const V4<T>* Tg = (const V4<T>*)A.Tg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        V4<T> F{0, 0, 0, 0}; T P = 0;
        if (A.M[i] == T(0)) {
            const int x = (int)(i % A.W), y = (int)(i / A.W);
            const V4<T> xc = X[i], tc = Tg[i];
            const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int nx = x + dx[n], ny = y + dy[n];
                if (nx < 0 || nx >= A.W || ny < 0 || ny >= A.H) continue;
                const long ni = (long)ny * A.W + nx;
                const V4<T> e = (xc - X[ni]) - (tc - Tg[ni]);
                F = F + (e + e);      // own residual (+1 * e) and the neighbour-centred one (-1 * -e)
                P += T(2);
            }
        }
        ((V4<T>*)r)[i] = V4<T>{-F.x, -F.y, -F.z, -F.w};
        ((V4<T>*)diag)[i] = V4<T>{P, P, P, P};
    }
}

template <class T, bool LM>
__global__ __launch_bounds__(kBlock) void poisson_applyJTJ(PArgs<T> A, const T* __restrict__ v, T* __restrict__ out, const T* __restrict__ CtC, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    const long N = (long)A.W * A.H;
    const V4<T>* P = (const V4<T>*)v;
    double acc = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        V4<T> o{0, 0, 0, 0};
        if (A.M[i] == T(0)) {
            const int x = (int)(i % A.W), y = (int)(i / A.W);
            const V4<T> pc = P[i];
            const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int nx = x + dx[n], ny = y + dy[n];
                if (nx < 0 || nx >= A.W || ny < 0 || ny >= A.H) continue;
                const V4<T> d = pc - P[(long)ny * A.W + nx];   // p is 0 on excluded neighbours
                o = o + (d + d);
            }
            if (LM) { const V4<T> c = ((const V4<T>*)CtC)[i]; o = o + V4<T>{c.x * pc.x, c.y * pc.y, c.z * pc.z, c.w * pc.w}; }
            acc += (double)dot4(pc, o);
        }
        ((V4<T>*)out)[i] = o;
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
}

// ---- the operators of the marching PCG iteration (stencil_march.h) -------------------------------------------------------------------------
// poisson: (J^T J p)(c) = 2 sum over in-bounds neighbours of (p_c - p_n), neighbour order (+1,0), (-1,0), (0,+1), (0,-1) as in poisson_applyJTJ
template <class T>
struct PoissonMarchOp {
    static constexpr int C = 4, kCoef = 0; static constexpr bool kMasked = true, kSplit31 = false;
    // variants that would spill are not instantiated (tests/test_kernel_resources.py): double4 pixels at 768 threads (156 B), on chip 8 rows x 8 waves (528 B) / LM 4 x 8 (52 B)
    static constexpr int kMaxBlock = sizeof(T) == 8 ? 512 : 768;
    template <int R, int WV, bool LM> static constexpr bool spills() { return sizeof(T) == 8 && WV == 8 && (LM ? R == 4 : R == 8); }
    using Vec = MVec<T, 4>;
    __device__ __forceinline__ Vec apply(const Vec& pc, const Vec& pl, const Vec& pr, const Vec& pu, const Vec& pd, bool hasL, bool hasR, bool hasU, bool hasD, const MVec<T, 1>&) const {
        Vec o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            T a = 0;
            if (hasR) { const T d = pc.v[c] - pr.v[c]; a = a + (d + d); }
            if (hasL) { const T d = pc.v[c] - pl.v[c]; a = a + (d + d); }
            if (hasD) { const T d = pc.v[c] - pd.v[c]; a = a + (d + d); }
            if (hasU) { const T d = pc.v[c] - pu.v[c]; a = a + (d + d); }
            o.v[c] = a;
        }
        return o;
    }
};
// laplacian: 0.2^2 p_c + sum over in-bounds neighbours of (p_c - p_n), order as in lap_applyJTJ
struct LaplacianMarchOp {
    static constexpr int C = 1, kCoef = 0; static constexpr bool kMasked = false, kSplit31 = false;
    static constexpr int kMaxBlock = 768;
    template <int R, int WV, bool LM> static constexpr bool spills() { return false; }
    using Vec = MVec<float, 1>;
    __device__ __forceinline__ Vec apply(const Vec& pc, const Vec& pl, const Vec& pr, const Vec& pu, const Vec& pd, bool hasL, bool hasR, bool hasU, bool hasD, const MVec<float, 1>&) const {
        float o = 0.2f * 0.2f * pc.v[0];
        if (hasR) o += pc.v[0] - pr.v[0];
        if (hasL) o += pc.v[0] - pl.v[0];
        if (hasD) o += pc.v[0] - pd.v[0];
        if (hasU) o += pc.v[0] - pu.v[0];
        return Vec{{o}};
    }
};
template <class T>
__global__ __launch_bounds__(kBlock) void poisson_flags(const T* __restrict__ M, uint8_t* __restrict__ flags, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) flags[i] = M[i] == T(0) ? 1 : 0;
}

// ---- block-local "patch" PCG (the reference's LDS-resident comparator solver, examples/poisson_image_editing/src/
// PatchSolverWarping.cu:67-241 with the Halton-shifted tiling of :208-209) ------------------------------------------------------------
// One workgroup = one PS x PS patch of the image, shifted by (ox, oy).  It linearises at the current X (the problem is linear: b = -J^T F),
// runs nPatchIters iterations of Jacobi-preconditioned CG on the patch's own sub-system with the search direction resident in LDS, and adds
// the patch's delta to X.  Everything outside the patch -- and every excluded pixel -- keeps delta = 0 for this launch (the LDS apron stays
// 0), i.e. one launch is one additive-Schwarz sweep over non-overlapping blocks; successive launches shift the tiling so that block borders move.
// MI355X shape: PS = 32 gives 1024-thread workgroups (16 waves; a wave covers two 512-byte image rows), the only LDS array is p with its apron
// (18.5 KB in float): x, t and the mask are read once from HBM/L2 to form b and never staged.  Dot products: DPP wave sums + one LDS slot per
// wave, every thread adds the <= 16 wave sums in the same order (2 + 1 barriers per inner iteration; no atomics, deterministic).
// Unlike the reference kernel, which updates X in place while neighbouring blocks may still be reading their apron from it (a benign race that
// makes its output order-dependent), a launch reads Xin and writes Xout; the host ping-pongs the two.
template <class T, int NW> __device__ __forceinline__ T patchSum(T v, T* slot) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    T s = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += slot[i];
    return s;
}

template <class T, int PS>
__global__ __launch_bounds__(PS * PS) void poisson_patchSolve(int W, int H, const T* __restrict__ Xin, T* __restrict__ Xout, const T* __restrict__ Tg,
                                                              const T* __restrict__ M, int ox, int oy, int nPatchIters) {
    constexpr int LW = PS + 2, NW = PS * PS / kWave;
    __shared__ V4<T> P[LW * LW];
    __shared__ T red[2][NW];
    const int tid = threadIdx.x, tx = tid % PS, ty = tid / PS;
    const int gx = blockIdx.x * PS + tx - ox, gy = blockIdx.y * PS + ty - oy;
    const bool inImage = gx >= 0 && gx < W && gy >= 0 && gy < H;
    for (int i = tid; i < LW * LW; i += PS * PS) P[i] = V4<T>{0, 0, 0, 0};
    const long c = (long)gy * W + gx;
    const V4<T>* X = (const V4<T>*)Xin; const V4<T>* Tv = (const V4<T>*)Tg;
    const bool active = inImage && M[c] == T(0);
    const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
    V4<T> xc{0, 0, 0, 0}, R{0, 0, 0, 0}, delta{0, 0, 0, 0}, AP{0, 0, 0, 0}, pc{0, 0, 0, 0};
    T pre = 0; bool nb[4] = {false, false, false, false};
    if (inImage) xc = X[c];
    if (active) {
        const V4<T> tc = Tv[c];
        T cnt = 0;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nx = gx + dx[n], ny = gy + dy[n];
            nb[n] = nx >= 0 && nx < W && ny >= 0 && ny < H;
            if (!nb[n]) continue;
            const long ni = (long)ny * W + nx;
            const V4<T> e = (xc - X[ni]) - (tc - Tv[ni]);
            R = R - (e + e);                                  // r = -J^T F: the residual centred here and the one centred at the neighbour
            cnt += T(2);
        }
        pre = cnt > T(0) ? T(1) / cnt : T(1);                 // Jacobi: 1 / diag(J^T J)
        pc = pre * R;
    }
    __syncthreads();                                          // p zeroed (apron included) before the patch writes its own entries
    const int l = (ty + 1) * LW + tx + 1;
    if (active) P[l] = pc;
    T rz = patchSum<T, NW>(active ? dot4(R, pc) : T(0), red[1]);   // its barrier also publishes p
    if (rz > T(0)) {                                          // uniform: a patch without active pixels (or already converged) has nothing to do
        for (int it = 0; it < nPatchIters; ++it) {
            T d = 0;
            if (active) {
                AP = V4<T>{0, 0, 0, 0};
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (!nb[n]) continue;
                    const V4<T> q = pc - P[l + dy[n] * LW + dx[n]];   // p is 0 on excluded pixels and outside the patch
                    AP = AP + (q + q);
                }
                d = dot4(pc, AP);
            }
            const T den = patchSum<T, NW>(d, red[0]);
            T bn = 0;
            V4<T> Z{0, 0, 0, 0};
            if (active) {
                const T alpha = den > T(0) ? rz / den : T(0);
                delta = delta + alpha * pc;
                R = R - alpha * AP;
                Z = pre * R;
                bn = dot4(Z, R);
            }
            const T rzNew = patchSum<T, NW>(bn, red[1]);
            if (active) {
                const T beta = rz > T(0) ? rzNew / rz : T(0);
                pc = Z + beta * pc;
                P[l] = pc;
            }
            rz = rzNew;
            __syncthreads();
        }
    }
    if (inImage) ((V4<T>*)Xout)[c] = xc + delta;
}

template <class T>
struct PoissonOps : EnergyOps<T> {
    PArgs<T> A{};
    int cus = 256;
    bool singleKernel = true;
    MarchLoop<T> march; uint8_t* flags = nullptr;      // bit 0: M == 0 (the pixel is an unknown), refreshed at every bind
    PoissonOps(const unsigned* dims) {
        A.W = (int)dims[0]; A.H = (int)dims[1];
        this->usePreconditioner = false;                       // poisson_image_editing.t:5
        this->addUnknown(0, (long)A.W * A.H, 4);
        int dev = 0; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (const char* e = getenv("OPT_AMD_POISSON_ONEKERNEL")) singleKernel = atoi(e) != 0;
        if (singleKernel) oc.template reserveFor<PoissonMarchOp<T>>(A.W, A.H, cus);
    }
    int grid() const { return (int)std::max<long>(1, std::min<long>(((long)A.W * A.H + kBlock - 1) / kBlock, std::min<long>(kMaxPartials, (long)cus * 8))); }
    void bind(void** p, LaunchCtx& ctx) override {
        A.X = (const T*)p[0]; A.Tg = (const T*)p[1]; A.M = (const T*)p[2];
        if (singleKernel) {
            const long n = (long)A.W * A.H;
            if (!flags) HIP_CHECK(hipMalloc((void**)&flags, (size_t)n));
            poisson_flags<T><<<grid(), kBlock, 0, ctx.stream>>>(A.M, flags, n);
        }
    }
    T* unknownPtr(int) const override { return const_cast<T*>(A.X); }
    void evalCost(Reduction& out, LaunchCtx& ctx) override { ScopedKernel k(ctx, "computeCost"); poisson_cost<T, 0><<<grid(), kBlock, 0, ctx.stream>>>(A, nullptr, out.partials); out.n = grid(); }
    void evalJTF(T* r, T* diag, LaunchCtx& ctx) override { ScopedKernel k(ctx, "PCGInit1"); poisson_evalJTF<T><<<grid(), kBlock, 0, ctx.stream>>>(A, r, diag); }
    void applyJTJ(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "PCGStep1");
        if (CtC) poisson_applyJTJ<T, true><<<grid(), kBlock, 0, ctx.stream>>>(A, v, out, CtC, dot ? dot->partials : nullptr);
        else poisson_applyJTJ<T, false><<<grid(), kBlock, 0, ctx.stream>>>(A, v, out, nullptr, dot ? dot->partials : nullptr);
        if (dot) dot->n = grid();
    }
    void evalModelCost(const T* delta, Reduction& out, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "computeModelCost"); poisson_cost<T, 1><<<grid(), kBlock, 0, ctx.stream>>>(A, delta, out.partials); out.n = grid();
    }
    bool pcgIteration(const PcgIterArgs<T>& a, LaunchCtx& ctx) override {
        if (!singleKernel || a.pre || a.CtC) return false;      // Gauss-Newton only: the Levenberg-Marquardt loop keeps the generic kernels
        return march.launch(PoissonMarchOp<T>{}, A.W, A.H, flags, cus, a, ctx);
    }
    const T* pcgFinish(const T*, T* delta, LaunchCtx& ctx) override { return march.finish(delta, 4L * A.W * A.H, cus, ctx); }
    bool deltaMovable() const override { return !this->slab.active; }      // (the march takes delta from its arguments at every launch: PcgSolver::deltaTrial)
    // ---- the whole Gauss-Newton linear solve on chip (stencil_onchip.h) ----
    OnchipMarch<T> oc;
    bool onChipWithoutPreconditioner() const override { return true; }
    bool pcgSolveOnChip(const T* r0, const T* p0, T* delta, int L, double* traceDev, const OnChipLm<T>* lm, LaunchCtx& ctx) override {
        if (!singleKernel || traceDev || this->slab.active) return false;
        return oc.solve(PoissonMarchOp<T>{}, A.W, A.H, flags, nullptr, r0, p0, delta, const_cast<T*>(A.X), L, cus, ctx, lm);
    }
    bool onChipFailed() override { return oc.failedNow(); }
    bool onChipFailedPeek() override { return oc.failedPeek(); }
    void onChipRearm(LaunchCtx& ctx) override { oc.rearm(ctx); }
    std::string describe(int L, bool lmv) override { return oc.template describe<PoissonMarchOp<T>>(A.W, A.H, cus, singleKernel ? L : 0, lmv, "march_pcgIter"); }
    // ---- patch solver: ping-pong between the caller's X and a scratch copy; patchFinish leaves the result in the caller's buffer
    T* scratchX = nullptr; bool inScratch = false;
    bool supportsPatch() const override { return true; }
    ~PoissonOps() override { if (scratchX) (void)hipFree(scratchX); if (flags) (void)hipFree(flags); }
    template <int PS> void launchPatch(const T* in, T* out, float fx, float fy, int nPatchIters, LaunchCtx& ctx) {
        const dim3 g((A.W + PS - 1) / PS + 1, (A.H + PS - 1) / PS + 1);     // one more block per axis for the shift
        poisson_patchSolve<T, PS><<<g, PS * PS, 0, ctx.stream>>>(A.W, A.H, in, out, A.Tg, A.M, (int)(fx * PS), (int)(fy * PS), nPatchIters);
    }
    bool patchIteration(float fx, float fy, int nPatchIters, int patchSize, LaunchCtx& ctx) override {
        if (patchSize != 16 && patchSize != 32) return false;
        if (!scratchX) HIP_CHECK(hipMalloc((void**)&scratchX, (size_t)A.W * A.H * 4 * sizeof(T)));
        const T* in = inScratch ? scratchX : A.X; T* out = inScratch ? const_cast<T*>(A.X) : scratchX;
        ScopedKernel k(ctx, "PCGIterationPatch");
        if (patchSize == 16) launchPatch<16>(in, out, fx, fy, nPatchIters, ctx); else launchPatch<32>(in, out, fx, fy, nPatchIters, ctx);
        inScratch = !inScratch;
        return true;
    }
    void patchFinish(LaunchCtx& ctx) override {
        if (inScratch) HIP_CHECK(hipMemcpyAsync(const_cast<T*>(A.X), scratchX, (size_t)A.W * A.H * 4 * sizeof(T), hipMemcpyDeviceToDevice, ctx.stream));
        inScratch = false;
    }
};

// ---- laplacian (tests/minimal, tests/create_delete_cycle): float only --------------------------------------------
struct LArgs { int W, H; const float* X; const float* Aim; };

template <int MODE>   // 0 cost, 1 model cost
__global__ __launch_bounds__(kBlock) void lap_cost(LArgs A, const float* __restrict__ delta, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    const long N = (long)A.W * A.H;
    double acc = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % A.W), y = (int)(i / A.W);
        const float xc = A.X[i], dc = MODE ? delta[i] : 0.f;
        float f = 0.2f * (xc - A.Aim[i]) + (MODE ? 0.2f * dc : 0.f);
        float e = f * f;
        if (x + 1 < A.W) { float r = xc - A.X[i + 1]; if (MODE) r += dc - delta[i + 1]; e += r * r; }
        if (y + 1 < A.H) { float r = xc - A.X[i + A.W]; if (MODE) r += dc - delta[i + A.W]; e += r * r; }
        acc += (double)(0.5f * e);
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
__global__ __launch_bounds__(kBlock) void lap_evalJTF(LArgs A, float* __restrict__ r, float* __restrict__ diag) {
    const long N = (long)A.W * A.H;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % A.W), y = (int)(i / A.W);
        const float xc = A.X[i];
        float F = 0.2f * (0.2f * (xc - A.Aim[i])), P = 0.2f * 0.2f;
        if (x + 1 < A.W) { F += xc - A.X[i + 1]; P += 1.f; }
        if (x >= 1) { F -= A.X[i - 1] - xc; P += 1.f; }
        if (y + 1 < A.H) { F += xc - A.X[i + A.W]; P += 1.f; }
        if (y >= 1) { F -= A.X[i - A.W] - xc; P += 1.f; }
        r[i] = -F; diag[i] = P;
    }
}
template <bool LM>
__global__ __launch_bounds__(kBlock) void lap_applyJTJ(LArgs A, const float* __restrict__ v, float* __restrict__ out, const float* __restrict__ CtC, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    const long N = (long)A.W * A.H;
    double acc = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % A.W), y = (int)(i / A.W);
        const float pc = v[i];
        float o = 0.2f * 0.2f * pc;
        if (x + 1 < A.W) o += pc - v[i + 1];
        if (x >= 1) o += pc - v[i - 1];
        if (y + 1 < A.H) o += pc - v[i + A.W];
        if (y >= 1) o += pc - v[i - A.W];
        if (LM) o += CtC[i] * pc;
        out[i] = o;
        acc += (double)(pc * o);
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
}

struct LaplacianOps : EnergyOps<float> {
    LArgs A{};
    int cus = 256;
    LaplacianOps(const unsigned* dims) {
        A.W = (int)dims[0]; A.H = (int)dims[1];
        this->usePreconditioner = false;                       // no UsePreconditioner call in laplacian.t (default o.t:214)
        this->addUnknown(0, (long)A.W * A.H, 1);
        int dev = 0; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        oc.template reserveFor<LaplacianMarchOp>(A.W, A.H, cus);
    }
    int grid() const { return (int)std::max<long>(1, std::min<long>(((long)A.W * A.H + kBlock - 1) / kBlock, std::min<long>(kMaxPartials, (long)cus * 8))); }
    void bind(void** p, LaunchCtx&) override { A.X = (const float*)p[0]; A.Aim = (const float*)p[1]; }
    float* unknownPtr(int) const override { return const_cast<float*>(A.X); }
    void evalCost(Reduction& out, LaunchCtx& ctx) override { ScopedKernel k(ctx, "computeCost"); lap_cost<0><<<grid(), kBlock, 0, ctx.stream>>>(A, nullptr, out.partials); out.n = grid(); }
    void evalJTF(float* r, float* diag, LaunchCtx& ctx) override { ScopedKernel k(ctx, "PCGInit1"); lap_evalJTF<<<grid(), kBlock, 0, ctx.stream>>>(A, r, diag); }
    void applyJTJ(const float* v, float* out, const float* CtC, Reduction* dot, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "PCGStep1");
        if (CtC) lap_applyJTJ<true><<<grid(), kBlock, 0, ctx.stream>>>(A, v, out, CtC, dot ? dot->partials : nullptr);
        else lap_applyJTJ<false><<<grid(), kBlock, 0, ctx.stream>>>(A, v, out, nullptr, dot ? dot->partials : nullptr);
        if (dot) dot->n = grid();
    }
    void evalModelCost(const float* delta, Reduction& out, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "computeModelCost"); lap_cost<1><<<grid(), kBlock, 0, ctx.stream>>>(A, delta, out.partials); out.n = grid();
    }
    MarchLoop<float> march;
    bool pcgIteration(const PcgIterArgs<float>& a, LaunchCtx& ctx) override {
        if (a.pre || a.CtC) return false;
        return march.launch(LaplacianMarchOp{}, A.W, A.H, nullptr, cus, a, ctx);
    }
    const float* pcgFinish(const float*, float* delta, LaunchCtx& ctx) override { return march.finish(delta, (long)A.W * A.H, cus, ctx); }
    bool deltaMovable() const override { return !this->slab.active; }
    OnchipMarch<float> oc;      // the whole Gauss-Newton linear solve on chip (stencil_onchip.h)
    bool onChipWithoutPreconditioner() const override { return true; }
    bool pcgSolveOnChip(const float* r0, const float* p0, float* delta, int L, double* traceDev, const OnChipLm<float>* lm, LaunchCtx& ctx) override {
        if (traceDev || this->slab.active) return false;
        return oc.solve(LaplacianMarchOp{}, A.W, A.H, nullptr, nullptr, r0, p0, delta, const_cast<float*>(A.X), L, cus, ctx, lm);
    }
    bool onChipFailed() override { return oc.failedNow(); }
    bool onChipFailedPeek() override { return oc.failedPeek(); }
    void onChipRearm(LaunchCtx& ctx) override { oc.rearm(ctx); }
    std::string describe(int L, bool lmv) override { return oc.describe<LaplacianMarchOp>(A.W, A.H, cus, L, lmv, "march_pcgIter"); }
};

template <class T> EnergyOps<T>* makePoisson(const unsigned* dims) { return new PoissonOps<T>(dims); }
EnergyOps<float>* makeLaplacian(const unsigned* dims) { return new LaplacianOps(dims); }
EnergyOps<double>* makeLaplacianD(const unsigned*) { return nullptr; }

}  // namespace

EnergyInfo poissonInfo() {
    EnergyInfo e;
    e.name = "poisson_image_editing"; e.nDims = 2; e.usePreconditioner = false; e.floatOnly = false; e.residualsPerElement = 16;   // 4 directions x 4 channels
    e.params = {{ParamDecl::kUnknown, "X", "opt_float4", 0}, {ParamDecl::kArray, "T", "opt_float4", 1}, {ParamDecl::kArray, "M", "opt_float", 2}};
    e.makeFloat = makePoisson<float>; e.makeDouble = makePoisson<double>;
    return e;
}
EnergyInfo laplacianInfo() {
    EnergyInfo e;
    e.name = "laplacian"; e.nDims = 2; e.usePreconditioner = false; e.floatOnly = true;
    e.params = {{ParamDecl::kUnknown, "X", "float", 0}, {ParamDecl::kArray, "A", "float", 1}};
    e.makeFloat = makeLaplacian; e.makeDouble = makeLaplacianD;
    return e;
}

}  // namespace optamd
