// image_warping: the kernels that run once per Gauss-Newton step (and the J^T J application used by probes, the LM residual reset and the
// reference-ordered three-kernel loop, OPT_AMD_ONEKERNEL=0).  Energy and derivation: energy_image_warping.hip.
//   one thread per pixel (row slabs, probes, LM):  iw_flags, iw_cossin, iw_cost, iw_evalJTF, iw_checkLattice, iw_modelCost, iw_zeroGhost
//   row-marching (single GPU):                      iw_bindMarch (flags + unit-lattice verdict), iw_jtfMarch (PCGInit1 + PCGInit1_Finish), iw_costMarch
//   row-marching stencil:                           iw_applyJTJ (PCGStep1, optionally with the previous PCGStep3 fused in)
//   flat passes:                                    iw_finishUpdate (last delta terms + PCGLinearUpdate), iw_axpyDeferred
#pragma once
#include "iw_device.h"

namespace optamd {
namespace {

// once per Init/Step: fold Mask / Constraints / global bounds into one byte per pixel
template <class T>
__global__ __launch_bounds__(kBlock) void iw_flags(IWArgs<T> A) {
    const long N = (long)A.W * A.H;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / A.W), x = (int)(i % A.W), gy = A.gy0 + y;
        uint8_t f = 0;
        if (gy >= 0 && gy < A.Hg && A.Mask[i] == T(0)) f |= kActive;                          // eq(Mask,0)  (image_warping.t:11,17)
        if (A.Constraints[2 * i] >= T(0) && A.Constraints[2 * i + 1] >= T(0)) f |= kFit;      // All(greatereq(C,0)) (:22)
        // how many regularisation residuals v(c,n) are on: with it the Jacobi preconditioner of the Offset part (and, on a
        // unit lattice, of the Angle part) is a function of this byte alone and need not be streamed (iw_pcgIter2, PRE == 3)
        int cnt = 0;
        const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
        for (int n = 0; n < 4; ++n) {
            const int nx = x + dx[n], ny = y + dy[n], ngy = A.gy0 + ny;
            if (nx >= 0 && nx < A.W && ny >= 0 && ny < A.H && ngy >= 0 && ngy < A.Hg && A.Mask[(long)ny * A.W + nx] == T(0)) ++cnt;
        }
        A.flags[i] = f | (uint8_t)(cnt << kCountShift);
    }
}
// once per Gauss-Newton iteration: (cos a, sin a)
template <class T>
__global__ __launch_bounds__(kBlock) void iw_cossin(IWArgs<T> A) {
    const long N = (long)A.W * A.H;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        T s, c; sincosT(A.Angle[i], &s, &c);
        ((V2<T>*)A.cs)[i] = V2<T>{c, s};
    }
}

template <class T> __device__ __forceinline__ bool ownedRow(const IWArgs<T>& A, int y) { return y >= A.yBegin && y < A.yEnd; }

// ---- cost -----------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(kBlock) void iw_cost(IWArgs<T> A, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    const long rows = A.yEnd - A.yBegin, N = rows * A.W;
    const V2<T>* O = (const V2<T>*)A.Offset; const V2<T>* U = (const V2<T>*)A.UrShape; const V2<T>* C = (const V2<T>*)A.Constraints;
    double acc = 0;
    for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < N; j += (long)gridDim.x * blockDim.x) {
        const int x = (int)(j % A.W), y = A.yBegin + (int)(j / A.W);
        const long i = (long)y * A.W + x;
        const uint8_t f = A.flags[i];
        if (!(f & kActive)) continue;   // excluded pixel: its residuals are not part of the cost (solver.t:583)
        T s, c; sincosT(A.Angle[i], &s, &c);
        const V2<T> o = O[i], u = U[i];
        T e = 0;
        const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nx = x + dx[n], ny = y + dy[n];
            if (nx < 0 || nx >= A.W || ny < 0 || ny >= A.H) continue;
            const long ni = (long)ny * A.W + nx;
            if (!(A.flags[ni] & kActive)) continue;
            const V2<T> on = O[ni], un = U[ni];
            const T ux = u.x - un.x, uy = u.y - un.y;
            const T ex = A.w_reg * ((o.x - on.x) - (c * ux - s * uy));
            const T ey = A.w_reg * ((o.y - on.y) - (s * ux + c * uy));
            e += ex * ex + ey * ey;
        }
        if (f & kFit) {
            const V2<T> cc = C[i];
            const T fx = A.w_fit * (o.x - cc.x), fy = A.w_fit * (o.y - cc.y);
            e += fx * fx + fy * fy;
        }
        acc += (double)(T(0.5) * e);
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// ---- evalJTF: r = -J^T F, diag = diag(J^T J) ---------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(kBlock) void iw_evalJTF(IWArgs<T> A, T* __restrict__ r, T* __restrict__ diag) {
    const long N = (long)A.W * A.H;
    const V2<T>* O = (const V2<T>*)A.Offset; const V2<T>* U = (const V2<T>*)A.UrShape; const V2<T>* C = (const V2<T>*)A.Constraints;
    const V2<T>* CS = (const V2<T>*)A.cs;
    V2<T>* rO = (V2<T>*)r; T* ra = r + 2 * N; V2<T>* dO = (V2<T>*)diag; T* da = diag + 2 * N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % A.W), y = (int)(i / A.W);
        const uint8_t f = A.flags[i];
        T Fx = 0, Fy = 0, Fa = 0, Pxy = 0, Pa = 0;
        if ((f & kActive) && ownedRow(A, y)) {
            const V2<T> o = O[i], u = U[i], cs = CS[i];
            const T w = A.w_reg;
            const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int nx = x + dx[n], ny = y + dy[n];
                if (nx < 0 || nx >= A.W || ny < 0 || ny >= A.H) continue;
                const long ni = (long)ny * A.W + nx;
                if (!(A.flags[ni] & kActive)) continue;
                const V2<T> on = O[ni], un = U[ni], csn = CS[ni];
                const T ux = u.x - un.x, uy = u.y - un.y;
                // residual centred here, towards n
                const T ex = w * ((o.x - on.x) - (cs.x * ux - cs.y * uy));
                const T ey = w * ((o.y - on.y) - (cs.y * ux + cs.x * uy));
                // residual centred at the neighbour, towards here: -(O_c - O_n) + R(a_n)(U_c - U_n)
                const T gx = w * ((on.x - o.x) + (csn.x * ux - csn.y * uy));
                const T gy = w * ((on.y - o.y) + (csn.y * ux + csn.x * uy));
                Fx += w * ex - w * gx; Fy += w * ey - w * gy;
                const T Dx = -cs.y * ux - cs.x * uy, Dy = cs.x * ux - cs.y * uy;   // R'(a)(U_c - U_n)
                Fa += -(w * Dx) * ex - (w * Dy) * ey;
                Pxy += w * w + w * w;
                Pa += (w * Dx) * (w * Dx) + (w * Dy) * (w * Dy);
            }
            if (f & kFit) {
                const V2<T> cc = C[i];
                Fx += A.w_fit * (A.w_fit * (o.x - cc.x)); Fy += A.w_fit * (A.w_fit * (o.y - cc.y));
                Pxy += A.w_fit * A.w_fit;
            }
        }
        rO[i] = V2<T>{-Fx, -Fy}; ra[i] = -Fa;
        dO[i] = V2<T>{Pxy, Pxy}; da[i] = Pa;
    }
}

// ---- applyJTJ (PCGStep1), optionally with the previous iteration's PCGStep3 fused in -----------------------------
// Row-marching stencil: see the header comment.  With FUSE the kernel first forms the new search direction
// p = z + beta p (reference PCGStep3, solverGPUGaussNewton.t:537-550) for every pixel it touches -- the
// rows it owns plus its two halo rows -- writes it for the owned rows into a SECOND p buffer (in-place
// would race with the neighbouring workgroup's halo reads), and applies J^T J to it.  That removes one
// kernel and the re-read of p per PCG iteration.
//
// Lane layout: a wave covers 64 consecutive pixels of a row but only its inner 62 lanes produce output;
// lanes 0 and 63 are the horizontal halo (neighbouring waves overlap by 2 pixels).  Left / right
// neighbours are then whole-wave DPP shifts of registers -- no LDS, no divergent edge loads -- at the price
// of 3 % redundant lanes.  Rows are fetched two ahead of use (raw registers, combined late) so that a
// wave always has a full row of loads in flight while it computes.
template <class T>
struct Px {
    T ox, oy, a;    // v at this pixel (Offset part, Angle part)
    T c, s;         // cos/sin of the pixel's angle
    T ux, uy;       // UrShape
    int f;          // flags (0 if the pixel does not exist)
};
template <class T, bool FUSE>
struct Raw {        // one pixel's loads, not yet combined (keeps the loads independent of any ALU work)
    V2<T> o, cs, u; T a;
    V2<T> zo; T za;
    int f, ok;      // raw flag byte; ok = the pixel exists (known without the load)
};

template <bool RIGHT, class T> __device__ __forceinline__ Px<T> dppShiftPx(const Px<T>& p) {
    Px<T> q;
    q.ox = dppShift<RIGHT>(p.ox); q.oy = dppShift<RIGHT>(p.oy); q.a = dppShift<RIGHT>(p.a); q.c = dppShift<RIGHT>(p.c); q.s = dppShift<RIGHT>(p.s);
    q.ux = dppShift<RIGHT>(p.ux); q.uy = dppShift<RIGHT>(p.uy); q.f = dppShift<RIGHT>(p.f);
    return q;
}

template <class T>
struct FuseArgs {            // the PCGStep3 inputs when fused (see k_step3 in solver.hip)
    const T* z; T* vNew;
    const double* bNumPartials; int nB;
    const double* aNumOld; double* aNumNext;
    // RESET (the tail of LM's split residual reset in the same pass, solver.hip k_step2SecondHalf): r = b - A v, z = M r, partial sums of r . z and of 1/2 v . (r + b)
    const T* resetB = nullptr; const T* resetPre = nullptr; T* resetR = nullptr; T* resetZ = nullptr; double* resetBNum = nullptr; double* resetQ = nullptr;
};

template <class T, bool FUSE>
__device__ __forceinline__ Raw<T, FUSE> iw_loadRaw(const IWArgs<T>& A, const V2<T>* __restrict__ vO, const T* __restrict__ va, const V2<T>* __restrict__ zO,
                                                   const T* __restrict__ za, bool xok, int x, int y) {
    // Branch-free: out-of-image pixels read a clamped (valid) address and get flag 0; every use of the other
    // fields is gated by the flag through selects, so their values never matter.
    Raw<T, FUSE> r;
    const bool ok = xok && y >= 0 && y < A.H;
    const long i = (long)min(max(y, 0), A.H - 1) * A.W + min(max(x, 0), A.W - 1);
    r.f = A.flags[i]; r.ok = ok;      // NOT `ok ? f : 0` here: any ALU op on a loaded value forces its s_waitcnt before the loop back-edge
    r.o = (vO)[i]; r.a = (va)[i]; r.cs = ((const V2<T>*)A.cs)[i]; r.u = ((const V2<T>*)A.UrShape)[i];
    if (FUSE) { r.zo = (zO)[i]; r.za = (za)[i]; } else { r.zo = V2<T>{0, 0}; r.za = 0; }
    return r;
}
template <class T, bool FUSE>
__device__ __forceinline__ Px<T> iw_combine(const Raw<T, FUSE>& r, T beta) {
    Px<T> p;
    p.ox = r.o.x; p.oy = r.o.y; p.a = r.a;
    if (FUSE) { p.ox = r.zo.x + beta * p.ox; p.oy = r.zo.y + beta * p.oy; p.a = r.za + beta * p.a; }   // PCGStep3
    p.c = regCopy(r.cs.x); p.s = regCopy(r.cs.y); p.ux = regCopy(r.u.x); p.uy = regCopy(r.u.y); p.f = r.ok ? r.f : 0;
    if (!FUSE) { p.ox = regCopy(p.ox); p.oy = regCopy(p.oy); p.a = regCopy(p.a); }
    return p;
}

// accumulate the two residuals shared by centre c and neighbour n (the one centred at c and the one centred at n)
template <class T>
__device__ __forceinline__ void iw_pair(const Px<T>& c, const Px<T>& n, T& accOx, T& accOy, T& accA) {
    const bool on = (n.f & kActive) != 0;                              // v(c,n); the centre's own flag is applied by the caller
    const T ux = c.ux - n.ux, uy = c.uy - n.uy;
    const T Dcx = -c.s * ux - c.c * uy, Dcy = c.c * ux - c.s * uy;     // R'(a_c)(U_c - U_n)
    const T Dnx = n.s * ux + n.c * uy, Dny = -n.c * ux + n.s * uy;     // R'(a_n)(U_n - U_c)
    const T jcx = (c.ox - n.ox) - Dcx * c.a, jcy = (c.oy - n.oy) - Dcy * c.a;   // J p of the residual centred at c  (/w)
    const T jnx = (n.ox - c.ox) - Dnx * n.a, jny = (n.oy - c.oy) - Dny * n.a;   // J p of the residual centred at n  (/w)
    accOx += on ? jcx - jnx : T(0); accOy += on ? jcy - jny : T(0);   // selects, not branches: the kernel stays straight-line
    accA -= on ? Dcx * jcx + Dcy * jcy : T(0);
}

constexpr int kSpan = kWave - 2;                    // output pixels per wave per row
constexpr int kStrip = (kBlock / kWave) * kSpan;    // output pixels per workgroup per row (248)

// RESET (LM, not FUSE): v = delta; instead of storing A v the pass finishes the split residual reset (solver.t:505-534) -- what k_step2SecondHalf would do in a second pass
// over five vectors.
template <class T, bool LM, bool FUSE, bool RESET = false>
__global__ __launch_bounds__(kBlock) void iw_applyJTJ(IWArgs<T> A, const T* __restrict__ v, T* __restrict__ out, const T* __restrict__ CtC,
                                                      double* __restrict__ partials, int rowsPerGroup, int gx, int gy, FuseArgs<T> F) {
    static_assert(!RESET || (LM && !FUSE), "the reset tail belongs to the LM loop's plain J^T J pass");
    __shared__ double scratch[kBlock / kWave + 1];
    const int bx = blockIdx.x % gx, by = blockIdx.x / gx;
    const bool idle = by >= gy;
    const long N = (long)A.W * A.H;
    const V2<T>* vO = (const V2<T>*)v; const T* va = v + 2 * N;
    const V2<T>* zO = (const V2<T>*)F.z; const T* za = F.z + 2 * N;
    V2<T>* nO = (V2<T>*)F.vNew; T* na = F.vNew + 2 * N;
    V2<T>* outO = (V2<T>*)out; T* outA = out + 2 * N;
    const V2<T>* bO = (const V2<T>*)F.resetB; const T* bA = RESET ? F.resetB + 2 * N : nullptr;
    const V2<T>* mO = (const V2<T>*)F.resetPre; const T* mA = RESET ? F.resetPre + 2 * N : nullptr;
    V2<T>* rsO = (V2<T>*)F.resetR; T* rsA = RESET ? F.resetR + 2 * N : nullptr; V2<T>* zO2 = (V2<T>*)F.resetZ; T* zA2 = RESET ? F.resetZ + 2 * N : nullptr;
    double accQ = 0;
    T beta = 0;
    if (FUSE) {   // solver.t:541-547
        const double bSum = sumPartials(F.bNumPartials, F.nB, scratch);
        const T rDotzNew = (T)bSum, rDotzOld = (T)F.aNumOld[0];
        beta = (rDotzOld > T(0)) ? rDotzNew / rDotzOld : T(0);
        if (blockIdx.x == 0 && threadIdx.x == 0) F.aNumNext[0] = bSum;   // alphaNumerator <- betaNumerator (:1091)
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int x = bx * kStrip + wave * kSpan + lane - 1;
    const bool xok = x >= 0 && x < A.W;
    const bool writer = xok && lane >= 1 && lane <= kSpan;       // inner lanes own their pixel; lanes 0 / 63 are halo
    const int yb = idle ? A.yEnd : A.yBegin + by * rowsPerGroup;
    const int ye = idle ? A.yEnd : min(yb + rowsPerGroup, A.yEnd);
    const T w2 = A.w_reg * A.w_reg, wf2 = A.w_fit * A.w_fit;
    double acc = 0;

    Px<T> up = iw_combine<T, FUSE>(iw_loadRaw<T, FUSE>(A, vO, va, zO, za, xok, x, yb - 1), beta);
    Px<T> cur = iw_combine<T, FUSE>(iw_loadRaw<T, FUSE>(A, vO, va, zO, za, xok, x, yb), beta);
    if (FUSE && writer && yb < ye) {
        const long i = (long)yb * A.W + x;
        nO[i] = V2<T>{cur.ox, cur.oy}; na[i] = cur.a;
        if (yb - 1 >= 0 && yb == A.yBegin) { const long j = i - A.W; nO[j] = V2<T>{up.ox, up.oy}; na[j] = up.a; }   // ghost row above (slab mode)
    }
    // one row: `rdn` holds the raw loads of row y+1 (issued one iteration earlier)
    auto row = [&](int y, const Raw<T, FUSE>& rdn, bool live) {
        const Px<T> dn = iw_combine<T, FUSE>(rdn, beta);
        const long i = (long)y * A.W + x;
        if (FUSE && writer && live && y + 1 < A.H && (y + 1 < ye || y + 1 == A.yEnd)) { const long j = i + A.W; nO[j] = V2<T>{dn.ox, dn.oy}; na[j] = dn.a; }
        const Px<T> lf = dppShiftPx<true>(cur), rt = dppShiftPx<false>(cur);
        V2<T> bo{0, 0}, mo{0, 0}; T ba = 0, ma = 0;
        if (RESET) {      // requested before the row's arithmetic, used behind it
            const long ir = (writer && live) ? i : 0;
            bo = bO[ir]; ba = bA[ir]; mo = mO[ir]; ma = mA[ir];
        }
        T ax = 0, ay = 0, aa = 0;
        iw_pair(cur, rt, ax, ay, aa);
        iw_pair(cur, lf, ax, ay, aa);
        iw_pair(cur, dn, ax, ay, aa);
        iw_pair(cur, up, ax, ay, aa);
        T rx = w2 * ax, ry = w2 * ay, ra = w2 * aa;
        const bool fit = (cur.f & kFit) != 0;
        rx += fit ? wf2 * cur.ox : T(0); ry += fit ? wf2 * cur.oy : T(0);
        if (LM) {
            const long ic = (writer && live) ? i : 0;
            const V2<T> cO = ((const V2<T>*)CtC)[ic];
            rx += cO.x * cur.ox; ry += cO.y * cur.oy; ra += CtC[2 * N + ic] * cur.a;
        }
        const bool act = (cur.f & kActive) != 0;       // excluded / non-existent centre: row of J^T J is 0 (solver.t:424)
        rx = act ? rx : T(0); ry = act ? ry : T(0); ra = act ? ra : T(0);
        if (RESET) {
            if (writer && live) {
                const T r0 = bo.x - rx, r1 = bo.y - ry, r2 = ba - ra;       // r = b - A delta
                const T z0 = mo.x * r0, z1 = mo.y * r1, z2 = ma * r2;       // z = M r
                rsO[i] = V2<T>{r0, r1}; rsA[i] = r2; zO2[i] = V2<T>{z0, z1}; zA2[i] = z2;
                acc += (double)(z0 * r0) + (double)(z1 * r1) + (double)(z2 * r2);
                accQ += (double)(T(0.5) * (cur.ox * (r0 + bo.x))) + (double)(T(0.5) * (cur.oy * (r1 + bo.y))) + (double)(T(0.5) * (cur.a * (r2 + ba)));
            }
        } else if (writer && live) {
            acc += (double)(cur.ox * rx + cur.oy * ry + cur.a * ra);
            outO[i] = V2<T>{rx, ry}; outA[i] = ra;
        }
        up = cur; cur = dn;
    };
    // Two rows per trip, no branch around a load: a load inside a conditional block makes the compiler drain the whole
    // queue (s_waitcnt vmcnt(0)) where the paths merge, which would serialise the prefetch.  An odd last row runs as a
    // predicated no-op (clamped addresses, nothing stored or summed).
    Raw<T, FUSE> rA = iw_loadRaw<T, FUSE>(A, vO, va, zO, za, xok, x, yb + 1), rB;
    for (int y = yb; y < ye; y += 2) {
        __syncthreads();   // keep the 4 waves of a strip on the same rows: their shared seam lines then hit L2
        rB = iw_loadRaw<T, FUSE>(A, vO, va, zO, za, xok, x, y + 2);
        row(y, rA, true);
        rA = iw_loadRaw<T, FUSE>(A, vO, va, zO, za, xok, x, y + 3);
        row(y + 1, rB, y + 1 < ye);
    }
    double t = blockReduceSum(acc, scratch);
    if (RESET) {
        if (threadIdx.x == 0) F.resetBNum[blockIdx.x] = t;
        __syncthreads();
        const double tq = blockReduceSum(accQ, scratch);
        if (threadIdx.x == 0) F.resetQ[blockIdx.x] = tq;
    } else if (threadIdx.x == 0 && partials) partials[blockIdx.x] = t;
}

// ---- the once-per-Gauss-Newton-step passes as row-marching kernels (round 3) ------------------------------------------------------
// iw_flags / iw_checkLattice / iw_cossin / iw_evalJTF / k_initFinish / iw_cost above are one-thread-per-pixel kernels that gather their four
// neighbours through L1 / L2: 0.22-0.39 of the HBM peak, together 0.93 ms per Gauss-Newton step at 4096^2 -- 1.4 % of a step of 400 PCG iterations but a
// third of a step with the reference's default of 10 (solverGPUGaussNewton.t:26-39).  The kernels below do the same work in the marching layout of
// iw_applyJTJ: a workgroup owns a 248-pixel column strip and a contiguous range of rows, a lane keeps rows y-1, y, y+1 of its column in registers, left
// and right neighbours are DPP shifts, every input row is fetched once:
//   iw_bindMarch  : flags (Mask, Constraints -> 1 byte) and the unit-lattice verdict of UrShape in one pass (21 B/px in, 1 out); the verdict goes to
//                   pinned host memory, nothing blocks;
//   iw_jtfMarch   : PCGInit1 + PCGInit1_Finish (solver.t:361-419): r = -J^T F, p = M r, sum r.p -- and, for a general UrShape, the compact Jacobi
//                   preconditioner {M_O, M_a} the iteration kernel reads; sincos inline, no (cos, sin) table, no diag / preconditioner vectors written;
//   iw_costMarch  : computeCost (solver.t:580-592).
// Each reproduces the expressions of the kernel it replaces term by term (same operands, same order), so the values are the same up to the order of
// the double partial sums.  Used on a single GPU; slabs keep the older kernels.
template <class T>
struct MPx {               // one pixel of the 3-row window
    T ox, oy;              // Offset
    T c, s;                // cos / sin of Angle
    T ux, uy;              // UrShape (dead on a unit lattice)
    int f;                 // flag byte; 0 if the pixel does not exist
};
template <bool RIGHT, bool LATTICE, class T> __device__ __forceinline__ MPx<T> dppShiftM(const MPx<T>& p) {
    MPx<T> q;
    q.ox = dppShift<RIGHT>(p.ox); q.oy = dppShift<RIGHT>(p.oy); q.c = dppShift<RIGHT>(p.c); q.s = dppShift<RIGHT>(p.s); q.f = dppShift<RIGHT>(p.f);
    if (LATTICE) { q.ux = 0; q.uy = 0; } else { q.ux = dppShift<RIGHT>(p.ux); q.uy = dppShift<RIGHT>(p.uy); }
    return q;
}
template <class T> struct MRaw { V2<T> o, u, cc; T a; int f, ok; };
template <class T, bool LATTICE, bool NEEDC>
__device__ __forceinline__ MRaw<T> iw_marchLoad(const IWArgs<T>& A, bool xok, int x, int y) {
    MRaw<T> r;
    r.ok = xok && y >= 0 && y < A.H;
    const long i = (long)min(max(y, 0), A.H - 1) * A.W + min(max(x, 0), A.W - 1);      // clamped: always a valid address, gated by r.ok
    r.f = A.flags[i];
    r.o = ((const V2<T>*)A.Offset)[i]; r.a = A.Angle[i];
    if (LATTICE) r.u = V2<T>{0, 0}; else r.u = ((const V2<T>*)A.UrShape)[i];
    if (NEEDC) r.cc = ((const V2<T>*)A.Constraints)[i]; else r.cc = V2<T>{0, 0};
    return r;
}
template <class T, bool LATTICE>
__device__ __forceinline__ MPx<T> iw_marchCombine(const MRaw<T>& r) {
    MPx<T> p;
    p.ox = r.o.x; p.oy = r.o.y;
    sincosT(r.a, &p.s, &p.c);                     // the same sincos as iw_cossin: the values the table would hold
    if (LATTICE) { p.ux = 0; p.uy = 0; } else { p.ux = r.u.x; p.uy = r.u.y; }
    p.f = r.ok ? r.f : 0;
    return p;
}
// workgroup -> (column strip, row range) as in iw_applyJTJ
struct MarchGeo { int x, yb, ye; bool xok, writer; };
template <class T>
__device__ __forceinline__ MarchGeo marchGeo(const IWArgs<T>& A, int rowsPerGroup, int gx, int gy) {
    MarchGeo g;
    const int bx = blockIdx.x % gx, by = blockIdx.x / gx;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    g.x = bx * kStrip + wave * kSpan + lane - 1;
    g.xok = g.x >= 0 && g.x < A.W;
    g.writer = g.xok && lane >= 1 && lane <= kSpan;
    g.yb = A.yBegin + by * rowsPerGroup;
    g.ye = min(g.yb + rowsPerGroup, A.yEnd);
    if (by >= gy) g.yb = g.ye = A.yEnd;
    return g;
}

// flags + lattice verdict.  notLattice: pinned host word, zeroed by the host before the launch; any workgroup that finds a violation stores 1.
template <class T, bool CHECK>
__global__ __launch_bounds__(kBlock) void iw_bindMarch(IWArgs<T> A, int* __restrict__ notLattice, int rowsPerGroup, int gx, int gy) {
    const MarchGeo g = marchGeo(A, rowsPerGroup, gx, gy);
    struct R { T m; V2<T> c, u; int ok; };
    auto load = [&](int y) {
        R r;
        r.ok = g.xok && y >= 0 && y < A.H && (A.gy0 + y) >= 0 && (A.gy0 + y) < A.Hg;      // the pixel exists in the (global) image
        const long i = (long)min(max(y, 0), A.H - 1) * A.W + min(max(g.x, 0), A.W - 1);
        r.m = A.Mask[i]; r.c = ((const V2<T>*)A.Constraints)[i];
        if (CHECK) r.u = ((const V2<T>*)A.UrShape)[i]; else r.u = V2<T>{0, 0};
        return r;
    };
    struct P { int act, fit, ok; T ux, uy; };
    auto combine = [&](const R& r) {
        P p;
        p.ok = r.ok; p.act = (r.ok && r.m == T(0)) ? 1 : 0;                               // eq(Mask,0)  (image_warping.t:11,17)
        p.fit = (r.c.x >= T(0) && r.c.y >= T(0)) ? 1 : 0;                                 // All(greatereq(C,0)) (:22)
        p.ux = r.u.x; p.uy = r.u.y;
        return p;
    };
    P up = combine(load(g.yb - 1)), cur = combine(load(g.yb));
    bool bad = false;
    auto row = [&](int y, const R& rdn, bool live) {
        const P dn = combine(rdn);
        const int aR = dppShift<false>(cur.act), aL = dppShift<true>(cur.act);
        const int cnt = aR + aL + dn.act + up.act;
        if (CHECK) {
            const T rx = dppShift<false>(cur.ux), ry = dppShift<false>(cur.uy);
            const int rok = dppShift<false>(cur.ok);
            if (g.writer && live && cur.ok) {
                if (rok && g.x + 1 < A.W) bad |= !(cur.ux - rx == T(-1) && cur.uy - ry == T(0));
                if (dn.ok) bad |= !(cur.ux - dn.ux == T(0) && cur.uy - dn.uy == T(-1));
            }
        }
        if (g.writer && live) A.flags[(long)y * A.W + g.x] = (uint8_t)((cur.act ? kActive : 0) | (cur.fit ? kFit : 0) | (cnt << kCountShift));
        up = cur; cur = dn;
    };
    R rA = load(g.yb + 1), rB;
    for (int y = g.yb; y < g.ye; y += 2) {
        __syncthreads();
        rB = load(y + 2);
        row(y, rA, true);
        rA = load(y + 3);
        row(y + 1, rB, y + 1 < g.ye);
    }
    if (CHECK && __any(bad) && (threadIdx.x & (kWave - 1)) == 0) __hip_atomic_store(notLattice, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// r = -J^T F, p = guardedInvert(diag J^T J) r, partial sums of r.p (the iteration kernels rebuild M themselves: no preconditioner vector is written)
// COST: the same pass also sums computeCost's 1/2 r^2 (iw_costMarch's expressions, operand for operand, on the same grid: the same partial sums) -- the end of one
// Gauss-Newton step and the PCGInit1 of the next read the same unknowns, so inside Opt_ProblemSolve the two marches are one (PcgSolver: costAndJTFInit).
// LMINIT: the pass is Levenberg-Marquardt's PCGInit1 + PCGSaveSSq + PCGFinalizeDiagonal (solver.t:361-419, 624-664; solver.hip k_finalizeDiagonal<T, true>, expression for
// expression, on the diagonal this pass has just formed): CtC, the LM preconditioner, b = r, p = M r, delta = 0 and -- first outer iteration -- SSq, instead of a cos / sin
// table pass, a gathering J^T F pass that parks diag J^T J in memory and a flat pass that reads it back.
template <class T>
struct JtfLm { T *CtC, *SSq, *delta, *pre, *b; T radius, minLm, maxLm; int saveSSq; double* qPartials; };
template <class T, bool LATTICE, bool COST, bool LMINIT = false>
__global__ __launch_bounds__(kBlock) void iw_jtfMarch(IWArgs<T> A, T* __restrict__ r, T* __restrict__ p, double* __restrict__ partials, double* __restrict__ costPartials,
                                                      int rowsPerGroup, int gx, int gy, JtfLm<T> L = JtfLm<T>{}) {
    __shared__ double scratch[kBlock / kWave + 1];
    const MarchGeo g = marchGeo(A, rowsPerGroup, gx, gy);
    const long N = (long)A.W * A.H;
    V2<T>* rO = (V2<T>*)r; T* ra = r + 2 * N; V2<T>* pO = (V2<T>*)p; T* pa = p + 2 * N;
    const T w = A.w_reg;
    double acc = 0, accCost = 0;
    T e = 0;
    MPx<T> up = iw_marchCombine<T, LATTICE>(iw_marchLoad<T, LATTICE, true>(A, g.xok, g.x, g.yb - 1));
    MRaw<T> rawCur = iw_marchLoad<T, LATTICE, true>(A, g.xok, g.x, g.yb);
    MPx<T> cur = iw_marchCombine<T, LATTICE>(rawCur);
    V2<T> ccCur = rawCur.cc;
    T Fx, Fy, Fa, Pxy, Pa;
    auto pair = [&](const MPx<T>& c, const MPx<T>& n, T dux, T duy) {       // iw_evalJTF's loop body for one direction; (dux, duy) = U_c - U_n on a unit lattice
        if (!(n.f & kActive)) return;
        const T ux = LATTICE ? dux : c.ux - n.ux, uy = LATTICE ? duy : c.uy - n.uy;
        const T ex = w * ((c.ox - n.ox) - (c.c * ux - c.s * uy));
        const T ey = w * ((c.oy - n.oy) - (c.s * ux + c.c * uy));
        const T hx = w * ((n.ox - c.ox) + (n.c * ux - n.s * uy));
        const T hy = w * ((n.oy - c.oy) + (n.s * ux + n.c * uy));
        if (COST) e += ex * ex + ey * ey;
        Fx += w * ex - w * hx; Fy += w * ey - w * hy;
        const T Dx = -c.s * ux - c.c * uy, Dy = c.c * ux - c.s * uy;
        Fa += -(w * Dx) * ex - (w * Dy) * ey;
        Pxy += w * w + w * w;
        Pa += (w * Dx) * (w * Dx) + (w * Dy) * (w * Dy);
    };
    auto row = [&](int y, const MRaw<T>& rdn, bool live) {
        const MPx<T> dn = iw_marchCombine<T, LATTICE>(rdn);
        const MPx<T> lf = dppShiftM<true, LATTICE>(cur), rt = dppShiftM<false, LATTICE>(cur);
        Fx = 0; Fy = 0; Fa = 0; Pxy = 0; Pa = 0; e = 0;
        if (cur.f & kActive) {
            pair(cur, rt, T(-1), T(0)); pair(cur, lf, T(1), T(0)); pair(cur, dn, T(0), T(-1)); pair(cur, up, T(0), T(1));
            if (cur.f & kFit) {
                const T fx = A.w_fit * (cur.ox - ccCur.x), fy = A.w_fit * (cur.oy - ccCur.y);
                if (COST) e += fx * fx + fy * fy;
                Fx += A.w_fit * fx; Fy += A.w_fit * fy;
                Pxy += A.w_fit * A.w_fit;
            }
        }
        if (g.writer && live) {
            if (COST) accCost += (double)(T(0.5) * e);
            const long i = (long)y * A.W + g.x;
            const T r0 = -Fx, r1 = -Fy, r2 = -Fa;
            const T sO = T(1) + sqrt(Pxy), sA = T(1) + sqrt(Pa);
            T mO = T(1) / (sO * sO), mA = T(1) / (sA * sA);         // solver.hip guardedInvert (solver.t:323-332)
            if (LMINIT) {
                T sso = mO, ssa = mA;      // PCGSaveSSq: the first outer iteration's guardedInvert(diag)
                V2<T>* ssO = (V2<T>*)L.SSq; T* ssA = L.SSq + 2 * N;
                if (L.saveSSq) { ssO[i] = V2<T>{sso, sso}; ssA[i] = ssa; } else { const V2<T> so = ssO[i]; sso = so.x; ssa = ssA[i]; }      // (diag's two Offset components are one value, so SSq's are too)
                const T invRadius = T(1) / L.radius;
                auto fin = [&](T diag, T S, T& c, T& m) {
                    const T unclamped = diag * invRadius;                 // computeCtC: diag(J^T J) / radius (o.t:2277-2279)
                    const T invS = T(1) / S, clampMul = invS / L.radius;
                    c = fmin(fmax(unclamped, L.minLm * clampMul), L.maxLm * clampMul);
                    m = T(1) / (c + L.radius * unclamped);
                };
                T cO, cA;
                fin(Pxy, sso, cO, mO); fin(Pa, ssa, cA, mA);
                ((V2<T>*)L.CtC)[i] = V2<T>{cO, cO}; L.CtC[2 * N + i] = cA;
                ((V2<T>*)L.pre)[i] = V2<T>{mO, mO}; L.pre[2 * N + i] = mA;
                ((V2<T>*)L.b)[i] = V2<T>{r0, r1}; L.b[2 * N + i] = r2;
                ((V2<T>*)L.delta)[i] = V2<T>{0, 0}; L.delta[2 * N + i] = 0;
                ((V2<T>*)A.cs)[i] = V2<T>{cur.c, cur.s};      // the cos / sin table the LM step's other passes read (model cost, the reset's J^T J pass): iw_cossin's values
            }
            const T p0 = mO * r0, p1 = mO * r1, p2 = mA * r2;
            rO[i] = V2<T>{r0, r1}; ra[i] = r2;
            pO[i] = V2<T>{p0, p1}; pa[i] = p2;
            acc += (double)(r0 * p0) + (double)(r1 * p1) + (double)(r2 * p2);
        }
        up = cur; cur = dn; ccCur = rdn.cc;
    };
    MRaw<T> rA = iw_marchLoad<T, LATTICE, true>(A, g.xok, g.x, g.yb + 1), rB;
    for (int y = g.yb; y < g.ye; y += 2) {
        __syncthreads();
        rB = iw_marchLoad<T, LATTICE, true>(A, g.xok, g.x, y + 2);
        row(y, rA, true);
        rA = iw_marchLoad<T, LATTICE, true>(A, g.xok, g.x, y + 3);
        row(y + 1, rB, y + 1 < g.ye);
    }
    const double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
    if (COST) {
        __syncthreads();
        const double tc = blockReduceSum(accCost, scratch);
        if (threadIdx.x == 0) costPartials[blockIdx.x] = tc;
    }
    if (LMINIT && threadIdx.x == 0) L.qPartials[blockIdx.x] = 0.0;      // Q_0 = 1/2 sum delta . (r + b) with delta = 0
}

// 1/2 sum r^2 over the non-excluded pixels of the workgroup's rows (iw_cost's expressions)
template <class T, bool LATTICE>
__global__ __launch_bounds__(kBlock) void iw_costMarch(IWArgs<T> A, double* __restrict__ partials, int rowsPerGroup, int gx, int gy) {
    __shared__ double scratch[kBlock / kWave + 1];
    const MarchGeo g = marchGeo(A, rowsPerGroup, gx, gy);
    double acc = 0;
    MPx<T> up = iw_marchCombine<T, LATTICE>(iw_marchLoad<T, LATTICE, true>(A, g.xok, g.x, g.yb - 1));
    MRaw<T> rawCur = iw_marchLoad<T, LATTICE, true>(A, g.xok, g.x, g.yb);
    MPx<T> cur = iw_marchCombine<T, LATTICE>(rawCur);
    V2<T> ccCur = rawCur.cc;
    T e;
    auto pair = [&](const MPx<T>& c, const MPx<T>& n, T dux, T duy) {
        if (!(n.f & kActive)) return;
        const T ux = LATTICE ? dux : c.ux - n.ux, uy = LATTICE ? duy : c.uy - n.uy;
        const T ex = A.w_reg * ((c.ox - n.ox) - (c.c * ux - c.s * uy));
        const T ey = A.w_reg * ((c.oy - n.oy) - (c.s * ux + c.c * uy));
        e += ex * ex + ey * ey;
    };
    auto row = [&](int y, const MRaw<T>& rdn, bool live) {
        const MPx<T> dn = iw_marchCombine<T, LATTICE>(rdn);
        const MPx<T> lf = dppShiftM<true, LATTICE>(cur), rt = dppShiftM<false, LATTICE>(cur);
        e = 0;
        if (cur.f & kActive) {       // excluded pixel: its residuals are not part of the cost (solver.t:583)
            pair(cur, rt, T(-1), T(0)); pair(cur, lf, T(1), T(0)); pair(cur, dn, T(0), T(-1)); pair(cur, up, T(0), T(1));
            if (cur.f & kFit) {
                const T fx = A.w_fit * (cur.ox - ccCur.x), fy = A.w_fit * (cur.oy - ccCur.y);
                e += fx * fx + fy * fy;
            }
        }
        if (g.writer && live) acc += (double)(T(0.5) * e);
        up = cur; cur = dn; ccCur = rdn.cc;
    };
    MRaw<T> rA = iw_marchLoad<T, LATTICE, true>(A, g.xok, g.x, g.yb + 1), rB;
    for (int y = g.yb; y < g.ye; y += 2) {
        __syncthreads();
        rB = iw_marchLoad<T, LATTICE, true>(A, g.xok, g.x, y + 2);
        row(y, rA, true);
        rA = iw_marchLoad<T, LATTICE, true>(A, g.xok, g.x, y + 3);
        row(y + 1, rB, y + 1 < g.ye);
    }
    const double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// End of a Gauss-Newton linear solve in one pass over the unknowns: the deferred term of the paired delta update (if owed), the last PCGStep2's
// delta += alpha p (solver.t:461-462) and PCGLinearUpdate X += delta (:552-557) -- X = X + ((delta [+ a2 p2]) + a1 p1), the reference's order of
// additions.  delta itself is dead after the update and is not written back.  a1 = alphaNum / alphaDen of the last launch (guarded like PCGStep2's).
template <class T>
__global__ __launch_bounds__(kBlock) void iw_finishUpdate(T* __restrict__ XO, T* __restrict__ XA, const T* __restrict__ delta, const T* __restrict__ p1, const T* __restrict__ p2,
                                                          const T* __restrict__ alpha2, long N, const double* __restrict__ aNumPartials, int nNum,
                                                          const double* __restrict__ aDenPartials, int nDen) {
    __shared__ double scratch[2 * (kBlock / kWave + 1)];
    const double* const ps[2] = {aNumPartials, aDenPartials}; const int ns[2] = {nNum, nDen}; double o2[2];
    sumPartialsN<2>(ps, ns, scratch, o2);
    const T aNum = (T)o2[0], aDen = (T)o2[1];
    const T a1 = (aDen > T(0)) ? aNum / aDen : T(0);
    const T a2 = p2 ? alpha2[0] : T(0);
    const V2<T>* dO = (const V2<T>*)delta; const T* dA = delta ? delta + 2 * N : nullptr;      // delta == nullptr: it stands for 0 (no launch of the loop has written it)
    const V2<T>* qO = (const V2<T>*)p1; const T* qA = p1 + 2 * N;
    const V2<T>* sO = (const V2<T>*)p2; const T* sA = p2 ? p2 + 2 * N : nullptr;
    V2<T>* xO = (V2<T>*)XO;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        V2<T> d{0, 0}; T da = 0;
        if (delta) { d = dO[i]; da = dA[i]; }
        const V2<T> q = qO[i]; const T qa = qA[i];
        const V2<T> x = xO[i]; const T xa = XA[i];
        if (p2) { const V2<T> s = sO[i]; const T sa = sA[i]; d.x = d.x + a2 * s.x; d.y = d.y + a2 * s.y; da = da + a2 * sa; }
        d.x = d.x + a1 * q.x; d.y = d.y + a1 * q.y; da = da + a1 * qa;
        xO[i] = V2<T>{x.x + d.x, x.y + d.y}; XA[i] = xa + da;
    }
}

// delta += alpha[0] * p over n scalars (the deferred term left over when the PCG loop ends on an odd launch)
template <class T>
__global__ __launch_bounds__(kBlock) void iw_axpyDeferred(T* __restrict__ delta, const T* __restrict__ p, const T* __restrict__ alpha, long n) {
    const T a = alpha[0];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) delta[i] = delta[i] + a * p[i];
}

// Is UrShape a unit lattice (U(x,y) - U(x+1,y) == (-1,0) and U(x,y) - U(x,y+1) == (0,-1) exactly)?  The reference
// example always passes the pixel grid itself (examples/image_warping/src/CombinedSolver.h:161-172); any other input
// clears the flag and the general kernel runs.  Checked at every bind because the caller may swap buffers.
template <class T>
__global__ __launch_bounds__(kBlock) void iw_checkLattice(IWArgs<T> A, int* __restrict__ notLattice) {
    const long N = (long)A.W * A.H;
    const V2<T>* U = (const V2<T>*)A.UrShape;
    bool bad = false;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % A.W), y = (int)(i / A.W), gy = A.gy0 + y;
        if (gy < 0 || gy >= A.Hg) continue;       // slab mode: ghost rows beyond the global image hold no data (their pixels are inactive, U there is never used)
        const V2<T> u = U[i];
        if (x + 1 < A.W) { const V2<T> n = U[i + 1]; bad |= !(u.x - n.x == T(-1) && u.y - n.y == T(0)); }
        if (y + 1 < A.H && gy + 1 < A.Hg) { const V2<T> n = U[i + A.W]; bad |= !(u.x - n.x == T(0) && u.y - n.y == T(-1)); }
    }
    if (__any(bad) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(notLattice, 1);
}

// ghost rows of `out` are zeroed so the flat streaming kernels see r = 0 / Ap = 0 there (energy.h contract)
template <class T>
__global__ __launch_bounds__(kBlock) void iw_zeroGhost(IWArgs<T> A, T* __restrict__ out) {
    const long N = (long)A.W * A.H;
    const int ghostRows[2] = {A.yBegin - 1, A.yEnd};
    for (int g = 0; g < 2; ++g) {
        const int y = ghostRows[g];
        if (y < 0 || y >= A.H) continue;
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < A.W; x += gridDim.x * blockDim.x) {
            const long i = (long)y * A.W + x;
            ((V2<T>*)out)[i] = V2<T>{0, 0}; out[2 * N + i] = 0;
        }
    }
}

// ---- modelcost (LM): 1/2 sum (F + J delta)^2 ------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(kBlock) void iw_modelCost(IWArgs<T> A, const T* __restrict__ delta, double* __restrict__ partials) {
    __shared__ double scratch[kBlock / kWave + 1];
    const long rows = A.yEnd - A.yBegin, NN = rows * A.W, N = (long)A.W * A.H;
    const V2<T>* O = (const V2<T>*)A.Offset; const V2<T>* U = (const V2<T>*)A.UrShape; const V2<T>* C = (const V2<T>*)A.Constraints;
    const V2<T>* CS = (const V2<T>*)A.cs; const V2<T>* dO = (const V2<T>*)delta; const T* da = delta + 2 * N;
    double acc = 0;
    for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < NN; j += (long)gridDim.x * blockDim.x) {
        const int x = (int)(j % A.W), y = A.yBegin + (int)(j / A.W);
        const long i = (long)y * A.W + x;
        const uint8_t f = A.flags[i];
        if (!(f & kActive)) continue;
        const V2<T> o = O[i], u = U[i], cs = CS[i], d = dO[i];
        const T dang = da[i], w = A.w_reg;
        T e = 0;
        const int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nx = x + dx[n], ny = y + dy[n];
            if (nx < 0 || nx >= A.W || ny < 0 || ny >= A.H) continue;
            const long ni = (long)ny * A.W + nx;
            if (!(A.flags[ni] & kActive)) continue;
            const V2<T> on = O[ni], un = U[ni], dn = dO[ni];
            const T ux = u.x - un.x, uy = u.y - un.y;
            const T Dx = -cs.y * ux - cs.x * uy, Dy = cs.x * ux - cs.y * uy;
            const T mx = w * ((o.x - on.x) - (cs.x * ux - cs.y * uy)) + (w * (d.x - dn.x) - (w * Dx) * dang);
            const T my = w * ((o.y - on.y) - (cs.y * ux + cs.x * uy)) + (w * (d.y - dn.y) - (w * Dy) * dang);
            e += mx * mx + my * my;
        }
        if (f & kFit) {
            const V2<T> cc = C[i];
            const T fx = A.w_fit * (o.x - cc.x) + A.w_fit * d.x, fy = A.w_fit * (o.y - cc.y) + A.w_fit * d.y;
            e += fx * fx + fy * fy;
        }
        acc += (double)(T(0.5) * e);
    }
    double t = blockReduceSum(acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}


}  // namespace
}  // namespace optamd
