// shape_from_shading: depth refinement under spherical-harmonics lighting (BASELINE config 3: double LM).
//
// Energy restated (reference examples/shape_from_shading/shape_from_shading.t:1-90), per pixel c = (i,j):
//   ComputedArray B_I(c) = [interior(c) & D_i>0 at c,(-1,0),(0,-1)] (B(c) - I(c)),  B = SH shading of the normal built
//                 from d0 = X(-1,0), d1 = X(0,0), d2 = X(0,-1);  I = Im/2 + (Im(-1,0) + Im(0,-1))/4            (:32-66)
//   ComputedArray valid(c) = all five depths > 0 & four |X_c - X_nb| < 0.01 & interior                         (:82-87)
//   E_p   = [D_i > 0] w_p (X - D_i)                                                                             (:73-74)
//   E_g_h = [interior] w_g (B_I(0,0) - B_I(1,0)) edgeMaskR ,  E_g_v likewise with (0,1) / edgeMaskC            (:77-80)
//   E_s   = [valid] w_s (4 P(0,0) - P(-1,0) - P(0,-1) - P(1,0) - P(0,1)),  P(u) = ((i_u-u_x)/f_x d, (j_u-u_y)/f_y d, d)  (:89-90)
//   Exclude(D_i <= 0); no UsePreconditioner call -> false.
// Opt re-evaluates the ComputedArrays and their per-unknown gradient images in a `precompute` kernel after
// every update / revert (o.t:1007-1040, 2387-2409; solver.t:1005, 1116, 1155) and differentiates residuals that
// read them through those stored gradients (o.t:913-925).  The same structure is kept here: sfs_precompute
// writes B_I, dB_I/d{d0,d1,d2} and valid; everything else is linear algebra on those images.
//
// Kernel structure.  The shading rows couple a pixel with a 2-pixel neighbourhood (5 unknowns per row, 15 rows touching each unknown), so J^T J p is evaluated
// as row values (J v)_r of the five coupled residual rows of every pixel followed by a gather  out(c) = sum over the <= 16 rows that touch X_c of dr/dX_c * q_r:
//   sfs_pcgMarch  : one whole PCG iteration per launch as a row march (registers + DPP); with JTF = true the same march is PCGInit1 (+ PCGFinalizeDiagonal for LM);
//   sfs_costMarch : cost / model cost on the same layout;   sfs_applyTiled : plain J^T J v through LDS tiles for probes and the LM residual reset.
// At the 1024^2 config every array involved sits in the 256 MB Infinity Cache: the kernels are bound by cache bandwidth and by their seams, not by HBM.
#include "energy.h"
#include <cstdint>

namespace optamd {
namespace {

template <class T>
struct SArgs {
    int W, H;
    T w_p, w_s, w_g, f_x, f_y, u_x, u_y, L[9];
    const T* X; const T* D_i; const T* Im; const uint8_t* mR; const uint8_t* mC;
    T *B_I, *g0, *g1, *g2, *valid;     // ComputedArrays + gradient images
    uint32_t* fl2;                     // fl | edgeMaskR << 8 | edgeMaskC << 16: one word per pixel for sfs_pcgMarch (one load, one register, one DPP move per neighbour)
    uint8_t* fl;                       // bit 0: D_i > 0 (the unknown is not excluded), bit 1: valid == 1 -- one byte for the marching iteration kernel instead of two doubles
};

// 3-partial forward-mode scalar for the precompute kernel (the chain rule through the normalised normal)
template <class T> struct D3 { T v, a, b, c; };
template <class T> __device__ __forceinline__ D3<T> operator+(D3<T> x, D3<T> y) { return {x.v + y.v, x.a + y.a, x.b + y.b, x.c + y.c}; }
template <class T> __device__ __forceinline__ D3<T> operator-(D3<T> x, D3<T> y) { return {x.v - y.v, x.a - y.a, x.b - y.b, x.c - y.c}; }
template <class T> __device__ __forceinline__ D3<T> operator-(D3<T> x) { return {-x.v, -x.a, -x.b, -x.c}; }
template <class T> __device__ __forceinline__ D3<T> operator*(D3<T> x, D3<T> y) { return {x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b, x.c * y.v + x.v * y.c}; }
template <class T> __device__ __forceinline__ D3<T> operator*(T s, D3<T> x) { return {s * x.v, s * x.a, s * x.b, s * x.c}; }
template <class T> __device__ __forceinline__ D3<T> operator*(D3<T> x, T s) { return s * x; }
template <class T> __device__ __forceinline__ D3<T> operator/(D3<T> x, T s) { return (T(1) / s) * x; }
template <class T> __device__ __forceinline__ D3<T> rsqrtD(D3<T> x) { const T r = T(1) / sqrt(x.v); const T k = T(-0.5) * r / x.v; return {r, k * x.a, k * x.b, k * x.c}; }

template <class T> __device__ __forceinline__ bool sfs_interior(const SArgs<T>& A, int x, int y) { return x >= 1 && x <= A.W - 2 && y >= 1 && y <= A.H - 2; }
template <class T> __device__ __forceinline__ bool sfs_dv(const SArgs<T>& A, int x, int y) { return x >= 0 && x < A.W && y >= 0 && y < A.H && A.D_i[(long)y * A.W + x] > T(0); }

template <class T>
__global__ __launch_bounds__(kBlock) void sfs_precompute(SArgs<T> A) {
    // Branch-free loads (round 3): a border pixel reads the addresses of the interior pixel (1,1) and masks the results, so the thirteen loads of a pixel are in flight
    // together instead of queued behind `if (interior)` / `if (D_i > 0 ...)` (31 -> see DESIGN.md 3.3); the arithmetic and its order are unchanged.
    const long N = (long)A.W * A.H;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < N; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % A.W), y = (int)(e / A.W);
        const bool in1 = sfs_interior(A, x, y);
        const long c = in1 ? e : (long)A.W + 1;      // (A.W >= 3 && A.H >= 3 whenever an interior pixel exists; otherwise the clamp below keeps the address valid)
        const long cc = min(max(c, (long)0), N - 1), cl = min(max(c - 1, (long)0), N - 1), cr = min(max(c + 1, (long)0), N - 1),
                   cu = min(max(c - A.W, (long)0), N - 1), cd = min(max(c + A.W, (long)0), N - 1);
        const T X1 = A.X[cc], X0 = A.X[cl], Xr = A.X[cr], X2 = A.X[cu], Xd = A.X[cd];
        const T D1 = A.D_i[cc], D0 = A.D_i[cl], Dr = A.D_i[cr], D2 = A.D_i[cu], Dd = A.D_i[cd];
        const T I1 = A.Im[cc], I0 = A.Im[cl], I2 = A.Im[cu];
        const T Dself = A.D_i[e];
        T bi = 0, g0 = 0, g1 = 0, g2 = 0, vl = 0;
        if (in1) {
            if (D0 > T(0) && D1 > T(0) && D2 > T(0)) {
                const D3<T> d0{X0, 1, 0, 0}, d1{X1, 0, 1, 0}, d2{X2, 0, 0, 1};
                const T i = (T)x, j = (T)y;
                D3<T> nx = (d2 * (d1 - d0)) / A.f_y;
                D3<T> ny = (d0 * (d1 - d2)) / A.f_x;
                D3<T> nz = (nx * (A.u_x - i)) / A.f_x + (ny * (A.u_y - j)) / A.f_y - (d0 * d2) / (A.f_x * A.f_y);
                const D3<T> sq = nx * nx + ny * ny + nz * nz;
                if (sq.v > T(0)) { const D3<T> inv = rsqrtD(sq); nx = inv * nx; ny = inv * ny; nz = inv * nz; }
                D3<T> B = A.L[1] * ny + A.L[2] * nz + A.L[3] * nx + A.L[4] * (nx * ny) + A.L[5] * (ny * nz) +
                          A.L[6] * (-(nx * nx) - ny * ny + T(2) * (nz * nz)) + A.L[7] * (nz * nx) + A.L[8] * (nx * nx - ny * ny);
                B.v += A.L[0];
                const T Iv = I1 * T(0.5) + T(0.25) * (I0 + I2);
                bi = B.v - Iv; g0 = B.a; g1 = B.b; g2 = B.c;
            }
            bool v = D1 > T(0) && D2 > T(0) && Dd > T(0) && D0 > T(0) && Dr > T(0);
            const T thr = T(0.01);
            v = v && fabs(X1 - X2) < thr && fabs(X1 - Xd) < thr && fabs(X1 - X0) < thr && fabs(X1 - Xr) < thr;
            vl = v ? T(1) : T(0);
        }
        A.B_I[e] = bi; A.g0[e] = g0; A.g1[e] = g1; A.g2[e] = g2; A.valid[e] = vl;
        const unsigned fl = (Dself > T(0) ? 1u : 0u) | (vl == T(1) ? 2u : 0u);
        A.fl[e] = (uint8_t)fl;
        A.fl2[e] = fl | ((unsigned)A.mR[e] << 8) | ((unsigned)A.mC[e] << 16);
    }
}

template <class T> __device__ __forceinline__ T coefK(const SArgs<T>& A, int k, int x, int y) {
    return k == 0 ? ((T)x - A.u_x) / A.f_x : k == 1 ? ((T)y - A.u_y) / A.f_y : T(1);
}

// J^T J v in one launch through LDS: a workgroup owns a 32 x 8 tile of pixels, stages v and the gradient / mask images of the tile
// plus a 2-pixel apron once, forms the five row values (J v)_r of every row centre in the tile + 1-pixel ring in LDS, and gathers from there
// (out(c) = sum over the <= 16 rows that touch X_c of dr/dX_c * (J v)_r, + CtC v): ~90 B/px of HBM traffic.
#ifndef SFS_TW
#define SFS_TW 32
#define SFS_TH 8
#endif
constexpr int kSfsTW = SFS_TW, kSfsTH = SFS_TH;
// It serves probes and the LM residual reset (computeAdelta); the PCG loop runs on sfs_pcgMarch below.
// Arguments of the one-launch-per-iteration kernel (energy.h PcgIterArgs; A p is kept in memory: recomputing it would need a 4-pixel ring).  Launch k first applies
// PCGStep2 and PCGStep3 of the previous iteration to every pixel it stages -- r_k = r - alpha Ap_{k-1}, z_k = r_k (this energy does not precondition: pre = 1 after
// the start, solver.t:467-472), p_k = r_k + beta p_{k-1} -- writes r_k, p_k, delta += alpha p_{k-1} (and the Q partial sums for LM) for its own rows, then applies
// J^T J (+ CtC) to p_k.  betaNumerator = sum r_k^2 comes from the previous launch's sums by expansion: rr - 2 alpha s2 + alpha^2 s3 with rr = sum r_{k-1}^2,
// s2 = r.Ap, s3 = Ap.Ap.  The start is the reference's: p_0 comes from memory (PCGInit1: r_0 / 4; LM: the Jacobi-preconditioned r_0 of PCGFinalizeDiagonal,
// solver.t:650-656) and alphaNumerator_0 = r_0 . p_0, so launch 0 also sums rr_0 = r_0 . r_0 for launch 1's expansion.
template <class T>
struct SIterK {
    const T *rOld, *ApOld, *pOld; T *rNew, *pNew; const T* delta; T* deltaOut; const T* b; double* q; unsigned qTag;
    int first, restart, rrFromPrivate;
    const double *aNumPrev, *aDenPrev, *s2Prev, *s3Prev, *rrPrev; int nNum, nDen, n2, n3, nRR;
    const double *betaNum, *betaDen; int nBetaNum, nBetaDen;
    double *aNum, *aDen, *s2, *s3, *rr;
};
template <class T, bool LM>
__global__ __launch_bounds__(kBlock) void sfs_applyTiled(SArgs<T> A, const T* __restrict__ v, T* __restrict__ out, const T* __restrict__ CtC, double* __restrict__ partials) {
    constexpr int TW = kSfsTW, TH = kSfsTH, VW = TW + 4, VH = TH + 4, QW = TW + 2, QH = TH + 2;
    static_assert(TW * TH == kBlock, "one thread per tile pixel");
    __shared__ double scratch[kBlock / kWave + 1];
    __shared__ T sv[VH][VW], s0[VH][VW], s1[VH][VW], s2[VH][VW];      // v, g0, g1, g2 on the tile + 2 apron (index [y+2][x+2])
    __shared__ T sq[5][QH][QW];                                       // gh, gv, s0..s2 row values on the tile + 1 ring (index [y+1][x+1])
    __shared__ uint8_t smr[QH][QW], smc[QH][QW], sok[QH][QW], svalid[QH][QW];
    // P(u) coefficients (i_u - u_x) / f_x and (j_u - u_y) / f_y of the tile's columns / rows: one division per column and row of the footprint instead of
    // ~12 double divisions (~35 instructions each) per pixel in the phases below -- the same quotients, computed once
    __shared__ T cxTab[VW], cyTab[VH];
    const int tilesX = (A.W + TW - 1) / TW, tilesY = (A.H + TH - 1) / TH, nTiles = tilesX * tilesY;
    double acc = 0;
    // Everything a tile needs from global memory is requested one tile AHEAD into registers (`Pre`): the loads of tile t + gridDim.x are in
    // flight while tile t goes through its three LDS phases, so a workgroup pays the global latency once instead of twice per tile (the
    // own-pixel inputs of the gather phase -- D_i, CtC, delta, b -- used to be a second dependent round trip at the end of every tile).
    constexpr int NV = (VW * VH + kBlock - 1) / kBlock, NQ = (QW * QH + kBlock - 1) / kBlock;      // staged pixels per thread: 2 and 2
    struct Pre { T a[NV], g0[NV], g1[NV], g2[NV]; T vl[NQ]; uint8_t mr[NQ], mc[NQ]; T Di, ctc; };
    auto fetch = [&](int t) {
        Pre P;
        const int x0 = (t % tilesX) * TW, y0 = (t / tilesX) * TH;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = min((int)threadIdx.x + j * kBlock, VW * VH - 1);
            const int lx = i % VW, ly = i / VW, gx = x0 + lx - 2, gy = y0 + ly - 2;
            const long g = (long)min(max(gy, 0), A.H - 1) * A.W + min(max(gx, 0), A.W - 1);      // clamped: no branch around a load; masked when staged
            P.a[j] = v[g];
            P.g0[j] = A.g0[g]; P.g1[j] = A.g1[g]; P.g2[j] = A.g2[g];
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int i = min((int)threadIdx.x + j * kBlock, QW * QH - 1);
            const int lx = i % QW, ly = i / QW, gx = x0 + lx - 1, gy = y0 + ly - 1;
            const long g = (long)min(max(gy, 0), A.H - 1) * A.W + min(max(gx, 0), A.W - 1);
            P.mr[j] = A.mR[g]; P.mc[j] = A.mC[g]; P.vl[j] = A.valid[g];
        }
        {
            const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
            const long e = (long)min(y0 + ty, A.H - 1) * A.W + min(x0 + tx, A.W - 1);
#ifndef SFS_LATE_OWN
#define SFS_LATE_OWN 1      // 1: the own-pixel inputs of the gather phase are loaded there instead of with the tile (8 fewer live registers; config 3: 42.1 against 42.9 ms); 0 with SFS_PREFETCH
#endif
            if (!SFS_LATE_OWN) { P.Di = A.D_i[e]; P.ctc = LM ? CtC[e] : T(0); }
            else { P.Di = 0; P.ctc = 0; }
        }
        return P;
    };
#ifndef SFS_PREFETCH
#define SFS_PREFETCH 0      // 1: request the next tile's global inputs one tile ahead (A/B: 44.1 ms against 42.7 for config 3 -- the 76 extra VGPRs cost a workgroup per CU)
#endif
    Pre cur = fetch(min((int)blockIdx.x, nTiles - 1));
    for (int t = blockIdx.x; t < nTiles; t += gridDim.x) {
        const int x0 = (t % tilesX) * TW, y0 = (t / tilesX) * TH;
#if SFS_PREFETCH
        const Pre nxt = fetch(min(t + (int)gridDim.x, nTiles - 1));   // the last tile is fetched once more instead of branching around the loads
#else
        if (t != (int)blockIdx.x) cur = fetch(t);
#endif
        __syncthreads();                                              // previous tile's readers are done
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = threadIdx.x + j * kBlock;
            if (i < VW * VH) {
                const int lx = i % VW, ly = i / VW, gx = x0 + lx - 2, gy = y0 + ly - 2;
                const bool in = gx >= 0 && gx < A.W && gy >= 0 && gy < A.H;
                sv[ly][lx] = in ? cur.a[j] : T(0);
                s0[ly][lx] = in ? cur.g0[j] : T(0); s1[ly][lx] = in ? cur.g1[j] : T(0); s2[ly][lx] = in ? cur.g2[j] : T(0);
            }
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int i = threadIdx.x + j * kBlock;
            if (i < QW * QH) {
                const int lx = i % QW, ly = i / QW, gx = x0 + lx - 1, gy = y0 + ly - 1;
                const bool ok = sfs_interior(A, gx, gy);
                sok[ly][lx] = ok; smr[ly][lx] = ok ? cur.mr[j] : 0; smc[ly][lx] = ok ? cur.mc[j] : 0; svalid[ly][lx] = ok && cur.vl[j] == T(1);
            }
        }
        if ((int)threadIdx.x < VW) cxTab[threadIdx.x] = coefK(A, 0, x0 + (int)threadIdx.x - 2, 0);
        else if ((int)threadIdx.x < VW + VH) cyTab[threadIdx.x - VW] = coefK(A, 1, 0, y0 + (int)threadIdx.x - VW - 2);
        __syncthreads();
        // row values at every centre of the tile + ring: centre (qx, qy) in q coordinates = (qx + 1, qy + 1) in v coordinates
        for (int i = threadIdx.x; i < QW * QH; i += kBlock) {
            const int qx = i % QW, qy = i / QW, vx = qx + 1, vy = qy + 1;
            // Branch-free: every LDS read below is inside the staged footprint for every (qx, qy); the masks (interior / valid) are applied as
            // selects at the end.  (Measured: no faster than the `if (interior) { ... if (valid) { ... } }` form -- 54.7 us per fused launch either way.)
            T jgh, jgv, js[3];
            {
                const bool ok = sok[qy][qx] != 0, vd = svalid[qy][qx] != 0;
                const T mr = (T)smr[qy][qx], mc = (T)smc[qy][qx];
                const T base = s1[vy][vx] * sv[vy][vx] + s0[vy][vx] * sv[vy][vx - 1] + s2[vy][vx] * sv[vy - 1][vx];
                const T right = s1[vy][vx + 1] * sv[vy][vx + 1] + s0[vy][vx + 1] * sv[vy][vx] + s2[vy][vx + 1] * sv[vy - 1][vx + 1];
                const T down = s1[vy + 1][vx] * sv[vy + 1][vx] + s0[vy + 1][vx] * sv[vy + 1][vx - 1] + s2[vy + 1][vx] * sv[vy][vx];
                jgh = ok ? A.w_g * mr * (base - right) : T(0); jgv = ok ? A.w_g * mc * (base - down) : T(0);
                const int ox[5] = {0, -1, 0, 1, 0}, oy[5] = {0, 0, -1, 0, 1};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    T sj = 0;
#pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        const T ck = k == 0 ? cxTab[vx + ox[u]] : k == 1 ? cyTab[vy + oy[u]] : T(1);      // coefK(A, k, gx + ox[u], gy + oy[u])
                        sj += ((u == 0 ? T(4) : T(-1)) * ck) * sv[vy + oy[u]][vx + ox[u]];
                    }
                    js[k] = (ok && vd) ? A.w_s * sj : T(0);
                }
            }
            sq[0][qy][qx] = jgh; sq[1][qy][qx] = jgv; sq[2][qy][qx] = js[0]; sq[3][qy][qx] = js[1]; sq[4][qy][qx] = js[2];
        }
        __syncthreads();
        // gather: this thread's pixel is (tx, ty) in the tile = (tx + 1, ty + 1) in q coordinates
        const int tx = threadIdx.x % TW, ty = threadIdx.x / TW, x = x0 + tx, y = y0 + ty;
        if (x < A.W && y < A.H) {
            const long e = (long)y * A.W + x;
            const T ve = sv[ty + 2][tx + 2];
            if (SFS_LATE_OWN) { cur.Di = A.D_i[e]; cur.ctc = LM ? CtC[e] : T(0); }
            T s = 0;
            {   // (branch-free like the row values above; an excluded pixel's sum is discarded by the select below)
                auto add = [&](T coef, T q) { s += coef * q; };
                add(A.w_p, A.w_p * ve);
                // (dx, dy): the row centre relative to this pixel; g arrays are read at the centre c and at c + 1 (gh) / c + W (gv)
                auto gh = [&](int dx, int dy, int slot) {
                    const int qx = tx + 1 + dx, qy = ty + 1 + dy, vx = qx + 1, vy = qy + 1;
                    const T m = A.w_g * (T)smr[qy][qx];
                    T coef = slot == 0 ? m * (s1[vy][vx] - s0[vy][vx + 1]) : slot == 1 ? m * s0[vy][vx] : slot == 2 ? m * s2[vy][vx] : slot == 3 ? -(m * s1[vy][vx + 1]) : -(m * s2[vy][vx + 1]);
                    coef = sok[qy][qx] ? coef : T(0);
                    add(coef, sq[0][qy][qx]);
                };
                auto gv = [&](int dx, int dy, int slot) {
                    const int qx = tx + 1 + dx, qy = ty + 1 + dy, vx = qx + 1, vy = qy + 1;
                    const T m = A.w_g * (T)smc[qy][qx];
                    T coef = slot == 0 ? m * (s1[vy][vx] - s2[vy + 1][vx]) : slot == 1 ? m * s0[vy][vx] : slot == 2 ? m * s2[vy][vx] : slot == 3 ? -(m * s1[vy + 1][vx]) : -(m * s0[vy + 1][vx]);
                    coef = sok[qy][qx] ? coef : T(0);
                    add(coef, sq[1][qy][qx]);
                };
                gh(0, 0, 0); gh(1, 0, 1); gh(0, 1, 2); gh(-1, 0, 3); gh(-1, 1, 4);
                gv(0, 0, 0); gv(1, 0, 1); gv(0, 1, 2); gv(0, -1, 3); gv(1, -1, 4);
                const int ox[5] = {0, 1, -1, 0, 0}, oy[5] = {0, 0, 0, 1, -1};
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int qx = tx + 1 + ox[u], qy = ty + 1 + oy[u];
                    const T wgt = svalid[qy][qx] ? A.w_s * (u == 0 ? T(4) : T(-1)) : T(0);
#pragma unroll
                    for (int k = 0; k < 3; ++k) add(wgt * (k == 0 ? cxTab[tx + 2] : k == 1 ? cyTab[ty + 2] : T(1)), sq[2 + k][qy][qx]);      // coefK(A, k, x, y)
                }
            }
            if (LM) s += cur.ctc * ve;
            if (!(cur.Di > T(0))) s = 0;
            out[e] = s;
            acc += (double)(ve * s);
        }
#if SFS_PREFETCH
        cur = nxt;
#endif
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
}

// ---- the PCG iteration as a ROW-MARCHING kernel ---------------------------------------------------------------------------------------------------------
// (An LDS-tiled form -- three barrier-separated phases per 32 x 8 tile, a 36 x 12 footprint staged for 256 outputs -- was latency-bound at 0.49 of the HBM peak and
// is gone since round 4.)  sfs_pcgMarch works in the layout of image_warping's iteration
// kernel: a wave owns 64 consecutive columns (the inner 60 are outputs; two halo lanes per side, because a pixel's gather reads row values on its 1-ring and
// those read p_k on their 1-ring) and marches down a range of rows, a lane keeping four staged rows of its column, two rows of dB.v and three rows of the five
// row values in registers.  Vertical neighbours cost nothing, horizontal ones are DPP wave shifts; there is no LDS staging and no barrier in the loop.
// Trip Y stages row Y (PCGStep2 + PCGStep3 of the previous iteration: r_k, p_k, and for the rows the workgroup owns the stores of r_k, p_k, delta and the Q
// sum), forms b(Y) = dB_I(., Y) . p_k, the five row values (J p_k)_r of the centres of row Y - 1, and the gather of row Y - 2 -- the expressions of
// sfs_applyTiled in the same order.
template <class T> struct SRaw { T r, p, ap, g0, g1, g2, ctc, dl, bb; int fb; };      // JTF mode: r = X, p = B_I, dl = D_i;  fb = SArgs::fl2 (flags and both edge masks in one word)
template <class T> struct SRow {
    T v, rk;               // p_k (what J^T J is applied to), r_k        (JTF mode: X, B_I)
    T g0, g1, g2, ctc;     // dB_I / d{d0, d1, d2} (0 outside the image), CtC
    int bits;              // kEx: D_i > 0 (the unknown is not excluded); kValid: interior row centre whose regularisation rows are on; kOk: interior row centre;
                           // bits 8-15 / 16-23: edge masks R / C, 0 unless the pixel is an interior row centre.  One register instead of five, one DPP move per neighbour.
};
constexpr int kSfsEx = 1, kSfsValid = 2, kSfsOk = 4;
__device__ __forceinline__ int sfsMr(int b) { return (b >> 8) & 255; }
__device__ __forceinline__ int sfsMc(int b) { return (b >> 16) & 255; }
template <class T> struct SQ { T gh, gv, s0, s1, s2; };
#ifndef SFS_MARCH_WAVES
#define SFS_MARCH_WAVES 4      // waves per workgroup (column strips side by side).  1024^2 double LM, us per iteration on one box: 1 wave 52.3, 2 36.3, 3 34.8, 4 33.9, 5 40.7 (1500 lanes for 1024 columns), 6 34.2, 8 34.0
#endif
constexpr int kSfsMarchBlock = SFS_MARCH_WAVES * kWave, kSfsSpan = kWave - 4;
#ifndef SFS_MARCH_MINWAVES
#define SFS_MARCH_MINWAVES 1      // waves per SIMD the register allocation must leave room for.  Round 4 took the double LM kernel from 215 to 183 VGPRs (flags and both edge masks in one
                                  // word, row coefficients in scalar registers, one accumulator for the two sums that never coexist); capped at 168 (3 waves per SIMD) it runs 45 us against 39:
                                  // at 1024^2 the grid is sized by co-residency, more resident workgroups mean fewer rows each, and every workgroup stages 4 halo rows (see marchGrid)
#endif
// JTF = true turns the same march into PCGInit1 (residual rows + gather in one launch): the staged vector is X itself, b is the stored B_I instead of dB_I . v, the
// row values are the residuals, the gather also sums the squared coefficients: out = -J^T F, diag = diag(J^T J).  No sums, no PCG state.
// With `fin.CtC` set the JTF march also does what k_finalizeDiagonal<T, true> does after PCGInit1 in an LM step (this energy: no preconditioner, no graph): SSq, delta = 0,
// the clamped CtC, the LM preconditioner, b = r, p = M r and the partial sums of r . p -- the same expressions on the value the gather has just produced.
template <class T> struct SFin { T *CtC, *SSq, *delta, *pre, *b, *p; T radius, minLm, maxLm; int saveSSq; double *dPart, *qPart; };
template <class T, bool LM, bool JTF = false>
__global__ __launch_bounds__(kSfsMarchBlock, (JTF ? 1 : SFS_MARCH_MINWAVES)) void sfs_pcgMarch(SArgs<T> A, T* __restrict__ out, const T* __restrict__ CtC, SIterK<T> K, int rowsPerGroup, int gx, int gy, int gyPerXcd,
                                                                                   T* __restrict__ diag = nullptr, SFin<T> fin = SFin<T>{}) {
    __shared__ double scratch[6 * (kSfsMarchBlock / kWave + 1)];
    T alpha = 0, beta = 0;
    const bool keep = JTF || K.first != 0 || K.restart != 0;       // r (and, at the start, p) are already those of this iteration
    // alpha, beta of this launch from the previous launch's partial sums -- called AFTER the first rows have been requested: their latency then overlaps this
    // memory round trip (partials another kernel just wrote) instead of following it
    auto prologue = [&]() {
        if (JTF) {
        } else if (K.restart) {
            const double* const ps[2] = {K.betaNum, K.betaDen}; const int ns[2] = {K.nBetaNum, K.nBetaDen}; double o2[2];
            sumPartialsN<2>(ps, ns, scratch, o2);
            const T bNum = (T)o2[0], bDen = (T)o2[1];
            beta = (bDen > T(0)) ? bNum / bDen : T(0);                 // solver.t:544-547
        } else if (!K.first) {
            const double* const ps[5] = {K.aNumPrev, K.aDenPrev, K.s2Prev, K.s3Prev, K.rrPrev}; const int ns[5] = {K.nNum, K.nDen, K.n2, K.n3, K.rrFromPrivate ? K.nRR : 0}; double o5[5];
            sumPartialsN<5>(ps, ns, scratch, o5);
            const T aNum = (T)o5[0], aDen = (T)o5[1];
            alpha = (aDen > T(0)) ? aNum / aDen : T(0);                // solver.t:456-459
            const double rr = K.rrFromPrivate ? o5[4] : o5[0];
            const double bNumD = fmax(rr - 2.0 * (double)alpha * o5[2] + (double)alpha * (double)alpha * o5[3], 0.0);
            beta = (aNum > T(0)) ? (T)bNumD / aNum : T(0);
        }
    };
    // workgroup -> (column strip, row group); the 8 XCDs take contiguous ranges of row groups, so that the halo rows two vertically adjacent workgroups both
    // stage are served by one L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int by = xcd * gyPerXcd + slot / gx, bx = slot % gx;
    const bool idle = by >= gy || slot / gx >= gyPerXcd;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int x = bx * (kSfsMarchBlock / kWave) * kSfsSpan + wave * kSfsSpan + lane - 2;
    const bool xin = x >= 0 && x < A.W;
    const bool writer = xin && lane >= 2 && lane < 2 + kSfsSpan;
    const int yb = idle ? A.H : by * rowsPerGroup, ye = idle ? A.H : min(yb + rowsPerGroup, A.H);
    const int xc = min(max(x, 0), A.W - 1);
    const T cxc = coefK(A, 0, x, 0);      // P(u) coefficient of this column; its neighbours' (the same expression at x - 1, x + 1) come from the neighbouring lanes where they are used
    double acc = 0, accNum = 0, acc2 = 0, acc3 = 0, accX = 0;      // accX: sum r^2 in the first launch of a solve (which applies no Step2), the Q sum in the others

    // Clamped addresses, masked when staged.  What only the workgroup's own rows need -- CtC, delta, b -- is read from row 0 on the halo rows (the same few cache
    // lines every time: L1 / L2 hits, no branch around a load): a halo row (4 of 14-23 per workgroup) costs 51 B per pixel of memory traffic instead of 75.
    auto load = [&](int y) {
        SRaw<T> w;
        const long g = (long)min(max(y, 0), A.H - 1) * A.W + xc;
        w.g0 = A.g0[g]; w.g1 = A.g1[g]; w.g2 = A.g2[g]; w.fb = (int)A.fl2[g];
        const long go = (y >= yb && y < ye) ? g : (long)xc;
        if (JTF) { w.r = A.X[g]; w.p = A.B_I[g]; w.ap = 0; w.ctc = 0; w.dl = A.D_i[go]; w.bb = 0; return w; }
        w.r = K.rOld[g]; w.p = K.pOld[g]; w.ap = K.ApOld[g];
        w.ctc = LM ? CtC[go] : T(0); w.dl = K.delta[go]; w.bb = LM ? K.b[go] : T(0);
        return w;
    };
    auto stage = [&](const SRaw<T>& w, int y) {
        SRow<T> n;
        const bool in = xin && y >= 0 && y < A.H;
        T rk = 0, pk = 0;
        if (JTF) { pk = in ? w.r : T(0); rk = in ? w.p : T(0); n.ctc = w.dl; }        // X, B_I; the own pixel's D_i rides in the ctc slot
        else if (in) {
            rk = keep ? w.r : w.r - alpha * w.ap;                                      // PCGStep2 (solver.t:464)
            pk = K.first ? w.p : rk + beta * w.p;                                      // PCGStep3 with z = r (solver.t:549)
        }
        n.v = pk; n.rk = rk;
        n.g0 = in ? w.g0 : T(0); n.g1 = in ? w.g1 : T(0); n.g2 = in ? w.g2 : T(0); if (!JTF) n.ctc = w.ctc;
        const bool ok = in && sfs_interior(A, x, y);
        n.bits = (in ? (w.fb & kSfsEx) : 0) | (ok ? ((w.fb & (kSfsValid | 0xffff00)) | kSfsOk) : 0);
        if (!JTF && writer && y >= yb && y < ye) {                                     // this workgroup's own rows
            const long g = (long)y * A.W + x;
            K.rNew[g] = rk; K.pNew[g] = pk;
            if (!keep) {                                                               // the rest of PCGStep2 of iteration k-1 for this pixel
                const T dl = w.dl + alpha * w.p;                                       // solver.t:461-462
                K.deltaOut[g] = dl;
                if (LM) accX += (double)(T(0.5) * (dl * (rk + w.bb)));                 // solver.t:483-485
            }
        }
        return n;
    };
    // the row coefficient is wave-uniform: kept in scalar registers (three live values, six VGPRs less)
    auto uni = [](T v) -> T {
        if constexpr (sizeof(T) == 8) return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
        else return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
    };
    auto cyOf = [&](int y) { return uni(coefK(A, 1, 0, y)); };

    // rows y+2, y+1, y, y-1 of the output row y; b = dB_I . v of rows y+2 / y+1; row values of rows y+1, y, y-1
    SRow<T> R1{}, R2{}, R3{};
    SQ<T> q2{}, q3{};
    T b1 = 0, cy1 = 0, cy2 = 0;
    auto trip = [&](int Y, const SRaw<T>& cur) {
        const SRow<T> n = stage(cur, Y);
        const T cyN = cyOf(Y);
        // b(., Y) = g1 v + g0 v(x-1) + g2 v(y-1)                                      (d B_I(c) . v)
        const T vL = dppShift<true>(n.v);
        const T bY = JTF ? n.rk : n.g1 * n.v + n.g0 * vL + n.g2 * R1.v;
        // row values at the centres of row Y - 1 (R1)
        SQ<T> qn;
        {
            const T right = dppShift<false>(b1);
            if (JTF) {      // residual values: w_g * ((B_I(c) - B_I(c + e)) * mask)
                qn.gh = (R1.bits & kSfsOk) ? A.w_g * ((b1 - right) * (T)sfsMr(R1.bits)) : T(0);
                qn.gv = (R1.bits & kSfsOk) ? A.w_g * ((b1 - bY) * (T)sfsMc(R1.bits)) : T(0);
            } else {
                qn.gh = (R1.bits & kSfsOk) ? A.w_g * (T)sfsMr(R1.bits) * (b1 - right) : T(0);
                qn.gv = (R1.bits & kSfsOk) ? A.w_g * (T)sfsMc(R1.bits) * (b1 - bY) : T(0);
            }
            const T v1l = dppShift<true>(R1.v), v1r = dppShift<false>(R1.v);
            const T cxl = dppShift<true>(cxc), cxr = dppShift<false>(cxc);      // (lanes 0 / 63 read 0: they are halo lanes whose row values are never used)
            T js[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const T c0 = k == 0 ? cxc : k == 1 ? cy1 : T(1), cl = k == 0 ? cxl : k == 1 ? cy1 : T(1), cu = k == 0 ? cxc : k == 1 ? cy2 : T(1),
                        cr = k == 0 ? cxr : k == 1 ? cy1 : T(1), cd = k == 0 ? cxc : k == 1 ? cyN : T(1);
                T sj = 0;
                sj += (T(4) * c0) * R1.v; sj += (T(-1) * cl) * v1l; sj += (T(-1) * cu) * R2.v; sj += (T(-1) * cr) * v1r; sj += (T(-1) * cd) * n.v;
                js[k] = (R1.bits & kSfsValid) ? A.w_s * sj : T(0);
            }
            qn.s0 = js[0]; qn.s1 = js[1]; qn.s2 = js[2];
        }
       
        // gather of row y = Y - 2 (centre row R2; row values qn at y + 1, q2 at y, q3 at y - 1)
        {
            const int y = Y - 2;
            const T ve = R2.v;
            T s = 0, dsum = 0;
            auto add = [&](T coef, T q) { s += coef * q; if (JTF) dsum += coef * coef; };
            add(A.w_p, JTF ? A.w_p * (ve - R2.ctc) : A.w_p * ve);      // the fitting row: w_p (X - D_i) / w_p v
            const T g0r = dppShift<false>(R2.g0);
            const int b2R = dppShift<false>(R2.bits), b2L = dppShift<true>(R2.bits), b1L = dppShift<true>(R1.bits), b3R = dppShift<false>(R3.bits);
            const int mrR = sfsMr(b2R), mrL = sfsMr(b2L), okR = b2R & kSfsOk, okL = b2L & kSfsOk;
            const int mr1L = sfsMr(b1L), ok1L = b1L & kSfsOk;
            const int mcR = sfsMc(b2R), mc3R = sfsMc(b3R), ok3R = b3R & kSfsOk;
            { const T m = A.w_g * (T)sfsMr(R2.bits); T coef = m * (R2.g1 - g0r); coef = (R2.bits & kSfsOk) ? coef : T(0); add(coef, q2.gh); }                       // gh, centre (x, y)
            { const T m = A.w_g * (T)mrR; T coef = m * g0r; coef = okR ? coef : T(0); add(coef, dppShift<false>(q2.gh)); }                     // (x+1, y)
            { const T m = A.w_g * (T)sfsMr(R1.bits); T coef = m * R1.g2; coef = (R1.bits & kSfsOk) ? coef : T(0); add(coef, qn.gh); }                                 // (x, y+1)
            { const T m = A.w_g * (T)mrL; T coef = -(m * R2.g1); coef = okL ? coef : T(0); add(coef, dppShift<true>(q2.gh)); }                  // (x-1, y)
            { const T m = A.w_g * (T)mr1L; T coef = -(m * R1.g2); coef = ok1L ? coef : T(0); add(coef, dppShift<true>(qn.gh)); }                // (x-1, y+1)
            { const T m = A.w_g * (T)sfsMc(R2.bits); T coef = m * (R2.g1 - R1.g2); coef = (R2.bits & kSfsOk) ? coef : T(0); add(coef, q2.gv); }                       // gv, centre (x, y)
            { const T m = A.w_g * (T)mcR; T coef = m * g0r; coef = okR ? coef : T(0); add(coef, dppShift<false>(q2.gv)); }                     // (x+1, y)
            { const T m = A.w_g * (T)sfsMc(R1.bits); T coef = m * R1.g2; coef = (R1.bits & kSfsOk) ? coef : T(0); add(coef, qn.gv); }                                 // (x, y+1)
            { const T m = A.w_g * (T)sfsMc(R3.bits); T coef = -(m * R2.g1); coef = (R3.bits & kSfsOk) ? coef : T(0); add(coef, q3.gv); }                              // (x, y-1)
            { const T m = A.w_g * (T)mc3R; T coef = -(m * g0r); coef = ok3R ? coef : T(0); add(coef, dppShift<false>(q3.gv)); }                 // (x+1, y-1)
            const int vR = b2R & kSfsValid, vLft = b2L & kSfsValid;
            auto reg = [&](int valid, T w4, T a0, T a1, T a2) {
                const T wgt = valid ? A.w_s * w4 : T(0);
                add(wgt * cxc, a0); add(wgt * cy2, a1); add(wgt * T(1), a2);
            };
            reg(R2.bits & kSfsValid, T(4), q2.s0, q2.s1, q2.s2);
            reg(vR, T(-1), dppShift<false>(q2.s0), dppShift<false>(q2.s1), dppShift<false>(q2.s2));
            reg(vLft, T(-1), dppShift<true>(q2.s0), dppShift<true>(q2.s1), dppShift<true>(q2.s2));
            reg(R1.bits & kSfsValid, T(-1), qn.s0, qn.s1, qn.s2);
            reg(R3.bits & kSfsValid, T(-1), q3.s0, q3.s1, q3.s2);
            if (JTF) {
                if (writer && y >= yb && y < ye) {
                    const long e = (long)y * A.W + x;
                    const T r0 = (R2.bits & kSfsEx) ? -s : -T(0), dg = (R2.bits & kSfsEx) ? dsum : T(0);
                    out[e] = r0;
                    if (fin.CtC) {      // k_finalizeDiagonal<T, true> with usePre = 0, graphMode = 0
                        const T s1 = T(1) + sqrt(T(1));
                        T S = T(1) / (s1 * s1);                               // guardedInvert(1)
                        if (fin.saveSSq) fin.SSq[e] = S; else S = fin.SSq[e];
                        fin.delta[e] = T(0);
                        const T invRadius = T(1) / fin.radius;
                        const T unclamped = dg * invRadius;                   // computeCtC: diag(J^T J) / radius (o.t:2277-2279)
                        const T invS = T(1) / S;
                        const T clampMul = invS / fin.radius;
                        const T lo = fin.minLm * clampMul, hi = fin.maxLm * clampMul;
                        const T c = fmin(fmax(unclamped, lo), hi);
                        fin.CtC[e] = c;
                        const T m = T(1) / (c + fin.radius * unclamped);
                        fin.pre[e] = m;
                        const T pp = m * r0;
                        fin.p[e] = pp; fin.b[e] = r0;
                        acc += (double)(r0 * pp);
                    } else diag[e] = dg;
                }
            } else {
            if (LM) s += R2.ctc * ve;
            if (!(R2.bits & kSfsEx)) s = 0;
            if (writer && y >= yb && y < ye) {
                out[(long)y * A.W + x] = s;
                acc += (double)(ve * s);
                const T rk = R2.rk;
                const T zk = K.first ? ve : rk;                                        // launch 0: alphaNumerator_0 = r_0 . p_0 (the reference's start)
                accNum += (double)(zk * rk); acc2 += (double)(rk * s); acc3 += (double)(s * s);
                if (K.first) accX += (double)(rk * rk);
            }
            }
        }
        R3 = R2; R2 = R1; R1 = n; q3 = q2; q2 = qn; b1 = bY; cy2 = cy1; cy1 = cyN;
    };
    // Named row buffers, each requested one (two) trips before it is consumed and consumed completely before it is requested again: no register copies of in-flight
    // loads at the back-edge (a copy of a just-requested buffer costs an s_waitcnt vmcnt(0) per trip; every wait in this loop is counted).
    // A trip count that is no multiple of the unroll runs past the end: clamped loads, nothing stored or summed (those rows are outside [yb, ye)).
#ifndef SFS_MARCH_DEPTH
#define SFS_MARCH_DEPTH 1      // rows in flight ahead of the one being worked on.  2 (three buffers, 237 VGPRs) measured no faster than 1 (two buffers, 215): 41.1-41.7 us per iteration against 39.4-40.0
#endif
    if (SFS_MARCH_DEPTH == 2) {
        SRaw<T> rA = load(yb - 2), rB = load(yb - 1), rC;
        prologue();
        for (int Y = yb - 2; Y < ye + 2; Y += 3) {
            rC = load(Y + 2); trip(Y, rA);
            rA = load(Y + 3); trip(Y + 1, rB);
            rB = load(Y + 4); trip(Y + 2, rC);
        }
    } else {
        SRaw<T> rA = load(yb - 2), rB;
        prologue();
        for (int Y = yb - 2; Y < ye + 2; Y += 2) {
            rB = load(Y + 1); trip(Y, rA);
            rA = load(Y + 2); trip(Y + 1, rB);
        }
    }
    if (JTF) {
        if (fin.CtC) {
            const double t = blockReduceSum(acc, scratch);
            if (threadIdx.x == 0) { fin.dPart[blockIdx.x] = t; fin.qPart[blockIdx.x] = 0.0; }
        }
        return;
    }
    double vv[6] = {acc, accNum, acc2, acc3, K.first ? accX : 0.0, K.first ? 0.0 : accX};
    blockReduceSumN<6>(vv, scratch);
    if (threadIdx.x == 0) {
        K.aDen[blockIdx.x] = vv[0]; K.aNum[blockIdx.x] = vv[1]; K.s2[blockIdx.x] = vv[2]; K.s3[blockIdx.x] = vv[3];
        if (K.first) K.rr[blockIdx.x] = vv[4];
        if (LM && K.q) { if (K.qTag) storeTaggedPartial(K.q, blockIdx.x, vv[5], K.qTag); else K.q[blockIdx.x] = vv[5]; }
    }
}

// ---- computeCost / computeModelCost as a march (round 3): the residual-row expressions of the energy (header), centre row Y - 1 from the rows Y - 2 .. Y of X (and of delta) held in registers ---------
template <class T, bool MODEL>
__global__ __launch_bounds__(kSfsMarchBlock) void sfs_costMarch(SArgs<T> A, const T* __restrict__ dl, double* __restrict__ partials, int rowsPerGroup, int gx, int gy, int gyPerXcd) {
    __shared__ double scratch[kSfsMarchBlock / kWave + 1];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int by = xcd * gyPerXcd + slot / gx, bx = slot % gx;
    const bool idle = by >= gy || slot / gx >= gyPerXcd;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int x = bx * (kSfsMarchBlock / kWave) * kSfsSpan + wave * kSfsSpan + lane - 2;
    const bool xin = x >= 0 && x < A.W;
    const bool writer = xin && lane >= 2 && lane < 2 + kSfsSpan;
    const int yb = idle ? A.H : by * rowsPerGroup, ye = idle ? A.H : min(yb + rowsPerGroup, A.H);
    const int xc = min(max(x, 0), A.W - 1);
    const T cxc = coefK(A, 0, x, 0), cxl = coefK(A, 0, x - 1, 0), cxr = coefK(A, 0, x + 1, 0);
    struct Raw { T X, B, Di, d, g0, g1, g2; int fl, mr, mc; };
    struct Row { T X, B, Di, d, db; int mr, mc, ok, valid, ex; };      // db = dB_I . delta of the row (MODEL)
    auto load = [&](int y) {
        Raw w;
        const long g = (long)min(max(y, 0), A.H - 1) * A.W + xc;
        w.X = A.X[g]; w.B = A.B_I[g]; w.Di = A.D_i[g]; w.fl = A.fl[g]; w.mr = A.mR[g]; w.mc = A.mC[g];
        if (MODEL) { w.d = dl[g]; w.g0 = A.g0[g]; w.g1 = A.g1[g]; w.g2 = A.g2[g]; } else { w.d = 0; w.g0 = 0; w.g1 = 0; w.g2 = 0; }
        return w;
    };
    double acc = 0;
    Row R1{}, R2{};
    T cy1 = 0, cy2 = 0;
    auto trip = [&](int Y, const Raw& w) {
        Row n;
        const bool in = xin && Y >= 0 && Y < A.H;
        n.X = in ? w.X : T(0); n.B = in ? w.B : T(0); n.Di = w.Di; n.d = in ? w.d : T(0);
        const bool ok = in && sfs_interior(A, x, Y);
        n.ok = ok; n.mr = ok ? w.mr : 0; n.mc = ok ? w.mc : 0; n.valid = ok && (w.fl & 2) != 0; n.ex = in && (w.fl & 1) != 0;
        const T cyN = coefK(A, 1, 0, Y);
        n.db = 0;
        if (MODEL) { const T dL = dppShift<true>(n.d); n.db = (in ? w.g1 : T(0)) * n.d + (in ? w.g0 : T(0)) * dL + (in ? w.g2 : T(0)) * R1.d; }      // d B_I(c) . delta
        // centre row Y - 1 (R1); rows Y - 2 (R2) and Y (n) around it
        {
            const int y = Y - 1;
            T rp = 0, jp = 0;
            if (R1.ex) { rp = A.w_p * (R1.X - R1.Di); if (MODEL) jp = A.w_p * R1.d; }
            const T bR = dppShift<false>(R1.B);
            T rgh = R1.ok ? A.w_g * ((R1.B - bR) * (T)R1.mr) : T(0), rgv = R1.ok ? A.w_g * ((R1.B - n.B) * (T)R1.mc) : T(0);
            T jgh = 0, jgv = 0;
            if (MODEL) {
                const T dbR = dppShift<false>(R1.db);
                jgh = R1.ok ? A.w_g * (T)R1.mr * (R1.db - dbR) : T(0); jgv = R1.ok ? A.w_g * (T)R1.mc * (R1.db - n.db) : T(0);
            }
            const T xl = dppShift<true>(R1.X), xr = dppShift<false>(R1.X);
            T dlf = 0, drt = 0;
            if (MODEL) { dlf = dppShift<true>(R1.d); drt = dppShift<false>(R1.d); }
            T rs[3], js[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const T c0 = k == 0 ? cxc : k == 1 ? cy1 : T(1), cl = k == 0 ? cxl : k == 1 ? cy1 : T(1), cu = k == 0 ? cxc : k == 1 ? cy2 : T(1),
                        cr = k == 0 ? cxr : k == 1 ? cy1 : T(1), cd = k == 0 ? cxc : k == 1 ? cyN : T(1);
                T sr = 0, sj = 0;
                sr += (T(4) * c0) * R1.X; sr += (T(-1) * cl) * xl; sr += (T(-1) * cu) * R2.X; sr += (T(-1) * cr) * xr; sr += (T(-1) * cd) * n.X;
                if (MODEL) { sj += (T(4) * c0) * R1.d; sj += (T(-1) * cl) * dlf; sj += (T(-1) * cu) * R2.d; sj += (T(-1) * cr) * drt; sj += (T(-1) * cd) * n.d; }
                rs[k] = R1.valid ? A.w_s * sr : T(0); js[k] = R1.valid ? A.w_s * sj : T(0);
            }
            if (R1.ex && writer && y >= yb && y < ye) {      // rows centred on excluded pixels are not part of the cost (solver.t:583, 669)
                const T a = rp + jp, b = rgh + jgh, c = rgv + jgv, d0 = rs[0] + js[0], d1 = rs[1] + js[1], d2 = rs[2] + js[2];
                acc += (double)(T(0.5) * (a * a + b * b + c * c + d0 * d0 + d1 * d1 + d2 * d2));
            }
        }
        R2 = R1; R1 = n; cy2 = cy1; cy1 = cyN;
    };
    Raw rA = load(yb - 1), rB;
    for (int Y = yb - 1; Y < ye + 1; Y += 2) {
        rB = load(Y + 1); trip(Y, rA);
        rA = load(Y + 2); trip(Y + 1, rB);
    }
    const double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

}  // namespace
}  // namespace optamd
#include "sfs_onchip.h"
namespace optamd {
namespace {

template <class T>
struct SfsOps : EnergyOps<T> {
    SArgs<T> A{};
    int cus = 256;
    std::vector<void*> owned;
    SfsOps(const unsigned* dims) {
        A.W = (int)dims[0]; A.H = (int)dims[1];
        this->usePreconditioner = false;                         // no UsePreconditioner call in the .t (default o.t:214)
        this->addUnknown(16, (long)A.W * A.H, 1);
        const size_t n = (size_t)A.W * A.H;
        T** imgs[5] = {&A.B_I, &A.g0, &A.g1, &A.g2, &A.valid};
        for (auto pp : imgs) { HIP_CHECK(hipMalloc((void**)pp, n * sizeof(T))); HIP_CHECK(hipMemset(*pp, 0, n * sizeof(T))); owned.push_back(*pp); }
        HIP_CHECK(hipMalloc((void**)&A.fl, n)); HIP_CHECK(hipMemset(A.fl, 0, n)); owned.push_back(A.fl);
        HIP_CHECK(hipMalloc((void**)&A.fl2, 4 * n)); HIP_CHECK(hipMemset(A.fl2, 0, 4 * n)); owned.push_back(A.fl2);
        int dev = 0; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (const char* e = getenv("OPT_AMD_SFS_ONEKERNEL")) oneKernel = atoi(e) != 0;
        gridOverride = devSwitch("OPT_AMD_SFS_GRID", gridOverride);
        if (const char* e = getenv("OPT_AMD_SFS_MARCH_GRID")) marchGridOverride = atoi(e);
        if (const char* e = getenv("OPT_AMD_ONCHIP")) soEnabled = atoi(e) != 0;
        if (const char* e = getenv("OPT_AMD_ONCHIP_ROWS")) soForceRows = std::max(0, atoi(e));
        if (const char* e = getenv("OPT_AMD_ONCHIP_WAVES")) soForceWaves = std::max(0, atoi(e));
        if (const char* e = getenv("OPT_AMD_ONCHIP_FAIL_AT")) soFailAt = atoi(e);      // test hook: see SfsOcArgs::failAt
        if (const char* e = getenv("OPT_AMD_ONCHIP_TIMEOUT_MS")) soTimeoutTicks = std::max(1, atoi(e)) * 100000LL;
        soReserve();
    }
    ~SfsOps() override { for (void* p : owned) (void)hipFree(p); if (soHostErr) (void)hipHostFree(soHostErr); }
    int grid() const { return (int)std::max<long>(1, std::min<long>(((long)A.W * A.H + kBlock - 1) / kBlock, std::min<long>(kMaxPartials, (long)cus * 8))); }
    void bind(void** p, LaunchCtx&) override {
        // sqrt(Param(...)) is evaluated in opt_float on the float parameter (shape_from_shading.t:4-6)
        A.w_p = std::sqrt((T) * (const float*)p[0]); A.w_s = std::sqrt((T) * (const float*)p[1]); A.w_g = std::sqrt((T) * (const float*)p[2]);
        A.f_x = (T) * (const float*)p[3]; A.f_y = (T) * (const float*)p[4]; A.u_x = (T) * (const float*)p[5]; A.u_y = (T) * (const float*)p[6];
        for (int i = 0; i < 9; ++i) A.L[i] = (T) * (const float*)p[7 + i];
        A.X = (const T*)p[16]; A.D_i = (const T*)p[17]; A.Im = (const T*)p[18]; A.mR = (const uint8_t*)p[19]; A.mC = (const uint8_t*)p[20];
    }
    T* unknownPtr(int) const override { return const_cast<T*>(A.X); }
    void precompute(LaunchCtx& ctx) override { ScopedKernel k(ctx, "precompute"); sfs_precompute<T><<<grid(), kBlock, 0, ctx.stream>>>(A); }
    void evalCost(Reduction& out, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "computeCost");
        int gx, gy, rows, per; marchGrid(false, gx, gy, rows, per);
        sfs_costMarch<T, false><<<8 * per * gx, kSfsMarchBlock, 0, ctx.stream>>>(A, nullptr, out.partials, rows, gx, gy, per); out.n = 8 * per * gx;
    }
    void evalJTF(T* r, T* diag, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "PCGInit1");      // row values + gather in one marching launch
        int gx, gy, rows, per; marchGrid(false, gx, gy, rows, per, true);
        sfs_pcgMarch<T, false, true><<<8 * per * gx, kSfsMarchBlock, 0, ctx.stream>>>(A, r, nullptr, SIterK<T>{}, rows, gx, gy, per, diag);
    }
    bool evalJTFInitLM(const LmInitArgs<T>& a, LaunchCtx& ctx) override {
        if (this->slab.active) return false;
        ScopedKernel k(ctx, "PCGInit1");
        int gx, gy, rows, per; marchGrid(false, gx, gy, rows, per, true);
        const int g = 8 * per * gx;
        SFin<T> fin{a.CtC, a.SSq, a.delta, a.pre, a.b, a.p, a.radius, a.minLm, a.maxLm, a.saveSSq, a.rDotP->partials, a.q->partials};
        sfs_pcgMarch<T, false, true><<<g, kSfsMarchBlock, 0, ctx.stream>>>(A, a.r, nullptr, SIterK<T>{}, rows, gx, gy, per, nullptr, fin);
        a.rDotP->n = g; a.q->n = g;
        return true;
    }
    // Grid of the tiled kernels: every workgroup loops over tiles, so the grid is capped at what is co-resident (LDS-limited: 4-5 workgroups per CU);
    // with more, the last round of workgroups runs on a partly empty chip (2048 workgroups on 1280 slots: 1.6 rounds).  (Development builds: OPT_AMD_SFS_GRID overrides.)
    int occTiled[2] = {0, 0}; int gridOverride = 0;
    int tileGrid(bool lmv) {
        const long t = (long)((A.W + kSfsTW - 1) / kSfsTW) * ((A.H + kSfsTH - 1) / kSfsTH);
        int& o = occTiled[lmv];
        if (o == 0) {
            const void* fn = lmv ? (const void*)sfs_applyTiled<T, true> : (const void*)sfs_applyTiled<T, false>;
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, fn, kBlock, 0));
            o = std::max(1, std::min(o, 8));
        }
        const long cap = gridOverride > 0 ? gridOverride : (long)cus * o;
        return (int)std::max<long>(1, std::min<long>(t, std::min<long>(kMaxPartials / 2, cap)));
    }
    void applyJTJ(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "PCGStep1");
        const int g = tileGrid(CtC != nullptr);
        if (CtC) sfs_applyTiled<T, true><<<g, kBlock, 0, ctx.stream>>>(A, v, out, CtC, dot ? dot->partials : nullptr);
        else sfs_applyTiled<T, false><<<g, kBlock, 0, ctx.stream>>>(A, v, out, nullptr, dot ? dot->partials : nullptr);
        if (dot) dot->n = g;
    }
    void evalModelCost(const T* delta, Reduction& out, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "computeModelCost");
        int gx, gy, rows, per; marchGrid(false, gx, gy, rows, per);
        sfs_costMarch<T, true><<<8 * per * gx, kSfsMarchBlock, 0, ctx.stream>>>(A, delta, out.partials, rows, gx, gy, per); out.n = 8 * per * gx;
    }
    // ---- one kernel per PCG iteration (sfs_pcgMarch) ----
    bool oneKernel = true;              // OPT_AMD_SFS_ONEKERNEL=0: the reference-ordered three kernels per iteration (the parity control)
    int occMarch[2] = {0, 0}, marchGridOverride = 0;
    // grid of the marching kernels: column strips of kSfsSpan columns per wave x row groups, 8 XCD-contiguous ranges of row groups
    void marchGrid(bool lmLoop, int& mgx, int& mgy, int& mRows, int& mPer, bool full = false) {
        int& o = occMarch[lmLoop];
        if (o == 0) {
            const void* fn = lmLoop ? (const void*)sfs_pcgMarch<T, true> : (const void*)sfs_pcgMarch<T, false>;
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, fn, kSfsMarchBlock, 0));
            o = std::max(1, std::min(o, 16));
        }
        mgx = divUp(A.W, (kSfsMarchBlock / kWave) * kSfsSpan);
        // Rows per workgroup against workgroups in flight: every workgroup stages 4 halo rows on top of its own (measured at 1024^2 double LM, us per iteration:
        // 1024 workgroups of 10 rows 44.0, 768 x 13 rows 41.6, 512 x 19 rows 42.1, 256 x 37 rows 59.3): the default takes three quarters of the co-resident count.
        // (PCGInit1 -- no prologue, more arithmetic per row -- prefers every co-resident slot: 30.5 us with 1024 workgroups against 33.8 with 768)
        // (4-wave workgroups, round 4: 256 / 320 / 384 / 448 / 512 / 576 workgroups 42.1 / 35.2 / 33.8 / 31.3-32.2 / 31.7-32.9 / 43.3 us, config 3 wall time best at 448: seven eighths)
        const int target = marchGridOverride > 0 ? marchGridOverride : std::max(1, full ? cus * o : cus * o * 7 / 8);
        mgy = std::max(1, std::min(std::min(A.H, target / mgx), (kMaxPartials / 2 - 8 * mgx) / mgx));
        mRows = divUp(A.H, mgy); mgy = divUp(A.H, mRows);
        mPer = divUp(mgy, 8);
    }
    double* rrPartials = nullptr; int nRR = 0; bool prevWasFirst = false;
    bool pcgIteration(const PcgIterArgs<T>& a, LaunchCtx& ctx) override {
        if (!oneKernel || a.pre || this->slab.active) return false;       // this energy does not precondition (pre == nullptr)
        if (a.ApNew == a.ApOld || a.rNew == a.rOld || a.pNew == a.pOld) return false;     // neighbouring tiles read the old apron while this one writes
        if (!rrPartials) { HIP_CHECK(hipMalloc((void**)&rrPartials, kMaxPartials * sizeof(double))); owned.push_back(rrPartials); }
        const bool lmLoop = a.CtC != nullptr;
        // the marching kernel's grid: column strips of kSfsSpan columns per wave x row groups sized to be co-resident, 8 XCD-contiguous ranges of row groups
        int mgx = 0, mgy = 0, mRows = 0, mPer = 0;
        marchGrid(lmLoop, mgx, mgy, mRows, mPer);
        const int g = 8 * mPer * mgx;
        SIterK<T> K{};
        K.rOld = a.rOld; K.ApOld = a.ApOld; K.pOld = a.pOld; K.rNew = a.rNew; K.pNew = a.pNew; K.delta = a.delta; K.deltaOut = a.deltaOut ? a.deltaOut : a.delta;
        K.b = a.b; K.q = a.q ? a.q->partials : nullptr; K.qTag = a.qTag; K.first = a.first; K.restart = a.afterReset;
        K.rrFromPrivate = (!a.first && !a.afterReset && prevWasFirst) ? 1 : 0;
        K.aNumPrev = a.aNumPrev.partials; K.aDenPrev = a.aDenPrev.partials; K.s2Prev = a.s2Prev.partials; K.s3Prev = a.s3Prev.partials; K.rrPrev = rrPartials;
        K.nNum = a.aNumPrev.n; K.nDen = a.aDenPrev.n; K.n2 = a.s2Prev.n; K.n3 = a.s3Prev.n; K.nRR = nRR;
        K.betaNum = a.betaNum.partials; K.betaDen = a.betaDen.partials; K.nBetaNum = a.betaNum.n; K.nBetaDen = a.betaDen.n;
        K.aNum = a.aNum->partials; K.aDen = a.aDen->partials; K.s2 = a.s2->partials; K.s3 = a.s3->partials; K.rr = rrPartials;
        {
            ScopedKernel k(ctx, "PCGIteration");
            if (lmLoop) sfs_pcgMarch<T, true><<<g, kSfsMarchBlock, 0, ctx.stream>>>(A, a.ApNew, a.CtC, K, mRows, mgx, mgy, mPer);
            else sfs_pcgMarch<T, false><<<g, kSfsMarchBlock, 0, ctx.stream>>>(A, a.ApNew, nullptr, K, mRows, mgx, mgy, mPer);
        }
        if (a.first) nRR = g;
        prevWasFirst = a.first != 0;
        a.aNum->n = a.aDen->n = a.s2->n = a.s3->n = g;
        if (a.q) a.q->n = g;
        return true;
    }

    // ---- the whole linear solve on chip (sfs_onchip.h): Gauss-Newton or Levenberg-Marquardt, one GPU, workgroups <= CUs ---------------------------------------------
    // OPT_AMD_ONCHIP=0 switches it off (the one A/B switch of the path); OPT_AMD_ONCHIP_ROWS=r / OPT_AMD_ONCHIP_WAVES=w force the variant that owns r rows per wave / has w waves per workgroup (tests run
    // every variant on small images); OPT_AMD_ONCHIP_FAIL_AT / _TIMEOUT_MS: the time-out path's test hooks.
    struct SoVariant { int rows, waves; const void* gn; const void* lm; };
    bool soEnabled = true, soFailed = false, soLaunched = false;
    int soForceRows = 0, soForceWaves = 0, soFailAt = -1; long long soTimeoutTicks = 0;      // 0: onchip_sync.h ocTimeouts() decides; OPT_AMD_ONCHIP_TIMEOUT_MS overrides
    long long* soProf = nullptr;
    oc_u64 *soSlots = nullptr, *soBox = nullptr; int *soBad = nullptr, *soHostErr = nullptr; unsigned soSeq = 0; size_t soSlotBytes = 0, soBoxBytes = 0;
    static const std::vector<SoVariant>& soVariants() {
        static const std::vector<SoVariant> v = [] {
            std::vector<SoVariant> o;
#define SO_VARIANT(R, WV) o.push_back({R, WV, (const void*)sfs_onchipPcg<T, R, false, WV>, (const void*)sfs_onchipPcg<T, R, true, WV>})
            SO_VARIANT(4, 4); SO_VARIANT(6, 4); SO_VARIANT(8, 4); SO_VARIANT(10, 4);
            SO_VARIANT(4, 8); SO_VARIANT(6, 8); SO_VARIANT(8, 8); SO_VARIANT(10, 8);
#undef SO_VARIANT
            return o;
        }();
        return v;
    }
    // Which variant, if any: among those whose workgroups fit one per CU, the one with the fewest marching trips per SIMD and iteration -- (waves per SIMD) x (rows
    // held per wave); ties go to the fewer rows (less work behind the wait).
    const SoVariant* soSelect(int& stripsX, int& tilesY, int& G) const {
        stripsX = divUp(A.W, kSoSpan);
        const SoVariant* best = nullptr; int bestCost = 1 << 30;
        for (const auto& v : soVariants()) {
            if (soForceRows && v.rows != soForceRows) continue;
            if (soForceWaves && v.waves != soForceWaves) continue;
            const int ty = divUp(A.H, v.rows), g = divUp(stripsX * ty, v.waves);
            if (g > std::min(cus, kSoMaxG)) continue;
            const int cost = (v.waves == 4 ? 100 : 136) * (v.rows + 4);      // (measured: two waves per SIMD march a pair of trips in 1.36 of the time one wave marches one)
            if (cost < bestCost) { best = &v; bestCost = cost; tilesY = ty; G = g; }
        }
        return best;
    }
    // the buffers of the path, sized for the plan's image when the plan is made (so that its first linear solve does not pay for the allocations); zero = no tag
    void soReserve() {
        if (soSlots || !soEnabled || (unsigned long long)A.W * A.H * sizeof(T) >= (1ull << 30)) return;
        { int sx, ty, g; if (!soSelect(sx, ty, g)) return; }      // (the image does not fit the chip: the path will never be taken)
        soSlotBytes = sizeof(oc_u64) * 2 * (size_t)kSoMaxG * kSoNW; soBoxBytes = sizeof(oc_u64) * 2 * (size_t)A.W * A.H * (sizeof(T) / 4);
        HIP_CHECK(hipMalloc((void**)&soSlots, soSlotBytes)); owned.push_back(soSlots);
        HIP_CHECK(hipMalloc((void**)&soBox, soBoxBytes)); owned.push_back(soBox);
        HIP_CHECK(hipMalloc((void**)&soBad, sizeof(int))); owned.push_back(soBad);
        HIP_CHECK(hipHostMalloc((void**)&soHostErr, 64)); *soHostErr = 0;
        HIP_CHECK(hipMemset(soBad, 0, sizeof(int))); HIP_CHECK(hipMemset(soSlots, 0, soSlotBytes)); HIP_CHECK(hipMemset(soBox, 0, soBoxBytes)); HIP_CHECK(hipStreamSynchronize(nullptr));      // (done before the plan's own stream sees the buffers)
        soSeq = 2;
#if SO_PROFILE
        if (getenv("OPT_AMD_ONCHIP_PROFILE")) { HIP_CHECK(hipMalloc((void**)&soProf, sizeof(long long) * 8 * kSoMaxG)); owned.push_back(soProf); }
#endif
    }
    bool onChipWithoutPreconditioner() const override { return true; }
    bool pcgSolveOnChip(const T* r0, const T* p0, T* delta, int L, double* traceDev, const OnChipLm<T>* lmArgs, LaunchCtx& ctx) override {
        if (!soEnabled || soFailed || this->slab.active || traceDev || L <= 0 || (unsigned long long)A.W * A.H * sizeof(T) >= (1ull << 30)) return false;
        if (lmArgs && (!lmArgs->CtC || lmArgs->resetPeriod < L)) return false;      // a split residual reset before the last iteration: the marching loop's business
        int stripsX = 0, tilesY = 0, G = 0;
        const SoVariant* V = soSelect(stripsX, tilesY, G);
        if (!V) return false;
        if (!soSlots) { soReserve(); if (!soSlots) return false; }
        if (soSeq > 0xE0000000u || soSeq + (unsigned)L > 0xE0000000u) {      // tags never repeat: start over on cleared buffers long before the counter wraps
            HIP_CHECK(hipMemsetAsync(soSlots, 0, soSlotBytes, ctx.stream)); HIP_CHECK(hipMemsetAsync(soBox, 0, soBoxBytes, ctx.stream));
            soSeq = 2;
        }
        const OcTimeouts tmo = ocTimeouts(soTimeoutTicks, L, false);
        SfsOcArgs<T> K{A, r0, p0, lmArgs ? lmArgs->CtC : nullptr, delta, stripsX, tilesY, G, L, soSeq, soSlots, soBox, soBad, tmo.later, soFailAt, tmo.first, lmArgs ? lmArgs->qTolerance : T(0), lmArgs ? soHostErr : nullptr, soProf, lmArgs ? lmArgs->breakInfo : nullptr};
        soSeq += (unsigned)L;
        {
            ScopedKernel k(ctx, "PCGSolveOnChip");
            void* kargs[] = {(void*)&K};
            if (hipLaunchKernel(lmArgs ? V->lm : V->gn, dim3(G), dim3(V->waves * kWave), kargs, 0, ctx.stream) != hipSuccess) {      // (a device that cannot hold the variant's LDS: not offered again)
                (void)hipGetLastError(); soEnabled = false; soSeq -= (unsigned)L;
                fprintf(stderr, "Opt(amd): the on-chip shape_from_shading kernel (%d rows, %d waves) could not be launched; the plan stays on the marching kernels\n", V->rows, V->waves);
                return false;
            }
        }
#if SO_PROFILE
        if (soProf) {      // development builds: where an iteration's time goes (thread 0 of every workgroup; mean and max over the workgroups)
            std::vector<long long> h((size_t)G * 8);
            HIP_CHECK(hipStreamSynchronize(ctx.stream));
            HIP_CHECK(hipMemcpy(h.data(), soProf, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            const char* names[6] = {"march", "wave-sums", "barrier", "wait", "grid-sum", "update"};
            fprintf(stderr, "sfs on-chip profile %dx%d rows=%d waves=%d G=%d L=%d (us per iteration: mean / max over workgroups):", A.W, A.H, V->rows, V->waves, G, L);
            for (int ph = 0; ph < 6; ++ph) {
                double mean = 0, mx = 0;
                for (int b = 0; b < G; ++b) { const double v = h[(size_t)b * 8 + ph] * 0.01 / L; mean += v / G; mx = std::max(mx, v); }
                fprintf(stderr, "  %s %.2f / %.2f", names[ph], mean, mx);
            }
            { double r6 = 0, r7 = 0; for (int b = 0; b < G; ++b) { r6 += (double)h[(size_t)b * 8 + 6] / L / G; r7 += h[(size_t)b * 8 + 7] * 0.01 / L / G; } fprintf(stderr, "  [thread 0: first round %.2f us, %.2f further rounds per iteration]", r7, r6); }
            fprintf(stderr, "\n");
        }
#endif
        if (!lmArgs) {      // (LM: the solver applies the update itself; a workgroup that gave up has told the host on its way out)
            ScopedKernel k(ctx, "PCGLinearUpdate");
            const long N = (long)A.W * A.H;
            sfs_applyDelta<T><<<grid(), kBlock, 0, ctx.stream>>>(const_cast<T*>(A.X), delta, N, soBad, soHostErr);
        }
        soLaunched = true;
        return true;
    }
    bool onChipFailed() override {
        if (!soLaunched) return false;
        soLaunched = false;
        if (__atomic_load_n(soHostErr, __ATOMIC_ACQUIRE) == 0) return false;
        soFailed = true;
        return true;
    }
    bool onChipFailedPeek() override { return soLaunched && soHostErr && __atomic_load_n(soHostErr, __ATOMIC_ACQUIRE) != 0; }
    void onChipRearm(LaunchCtx& ctx) override {
        if (!soBad) return;
        soFailed = false; __atomic_store_n(soHostErr, 0, __ATOMIC_RELEASE);
        HIP_CHECK(hipMemsetAsync(soBad, 0, sizeof(int), ctx.stream));
    }
    std::string describe(int L, bool lmv) override {      // ("key=value; ..." -- no ';' inside a value)
        int stripsX = 0, tilesY = 0, G = 0;
        const SoVariant* V = (soEnabled && !soFailed && !this->slab.active && L > 0) ? soSelect(stripsX, tilesY, G) : nullptr;
        char buf[600];
        if (V) snprintf(buf, sizeof buf, "path=on-chip (sfs_onchipPcg%s%s); onchip_rows_per_wave=%d; waves_per_workgroup=%d; wave_tiles=%dx%d of 60 x %d pixels; workgroups=%d of %d CUs; fallback=one launch per PCG iteration (sfs_pcgMarch)",
                        lmv ? ", LM" : "", lmv ? " while lIterations <= residual_reset_period" : "", V->rows, V->waves, stripsX, tilesY, V->rows, G, cus);
        else snprintf(buf, sizeof buf, "path=one launch per PCG iteration (sfs_pcgMarch%s); why_not_on_chip=%s", lmv ? ", LM" : "",
                      !soEnabled ? "switched off" : soFailed ? "a wait timed out earlier" : this->slab.active ? "row slabs" : "the wave tiles do not fit the CUs");
        return buf;
    }
};

template <class T> EnergyOps<T>* makeSfs(const unsigned* dims) { return new SfsOps<T>(dims); }

}  // namespace

EnergyInfo sfsInfo() {
    EnergyInfo e;
    e.name = "shape_from_shading"; e.nDims = 2; e.usePreconditioner = false; e.floatOnly = false; e.residualsPerElement = 6;    // E_p, E_g_h, E_g_v, E_s (3)
    e.params = {{ParamDecl::kScalar, "w_p", "float", 0}, {ParamDecl::kScalar, "w_s", "float", 1}, {ParamDecl::kScalar, "w_g", "float", 2},
                {ParamDecl::kScalar, "f_x", "float", 3}, {ParamDecl::kScalar, "f_y", "float", 4}, {ParamDecl::kScalar, "u_x", "float", 5},
                {ParamDecl::kScalar, "u_y", "float", 6},
                {ParamDecl::kScalar, "L_1", "float", 7}, {ParamDecl::kScalar, "L_2", "float", 8}, {ParamDecl::kScalar, "L_3", "float", 9},
                {ParamDecl::kScalar, "L_4", "float", 10}, {ParamDecl::kScalar, "L_5", "float", 11}, {ParamDecl::kScalar, "L_6", "float", 12},
                {ParamDecl::kScalar, "L_7", "float", 13}, {ParamDecl::kScalar, "L_8", "float", 14}, {ParamDecl::kScalar, "L_9", "float", 15},
                {ParamDecl::kUnknown, "X", "opt_float", 16}, {ParamDecl::kArray, "D_i", "opt_float", 17}, {ParamDecl::kArray, "Im", "opt_float", 18},
                {ParamDecl::kArray, "edgeMaskR", "uint8", 19}, {ParamDecl::kArray, "edgeMaskC", "uint8", 20}};
    e.makeFloat = makeSfs<float>; e.makeDouble = makeSfs<double>;
    return e;
}

}  // namespace optamd
