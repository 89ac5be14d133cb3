// image_warping: one whole PCG iteration per launch -- the streaming path for images that do not fit the chip (iw_onchip.h keeps smaller ones on chip).
//
// The reference runs PCGStep1 -> sum -> PCGStep2 -> sum -> PCGStep3 per iteration (solverGPUGaussNewton.t:1056-1092: three kernels, 180 B/pixel).  Launch k of
// iw_pcgIter2 does, for every pixel it touches,  r_k = r_{k-1} - alpha_{k-1} Ap_{k-1};  z_k = M r_k;  p_k = z_k + beta_{k-1} p_{k-1}  (Step2 + Step3 of
// iteration k-1), then  delta += alpha p  and  Ap_k = A p_k  on its own rows with the sums alphaDen = p.Ap, alphaNum = sum M r^2, s2 = sum M r.Ap, s3 = sum M Ap^2
// (beta by expansion: energy.h PcgIterArgs).  What keeps it at 53 B/pixel:
//   * A p is never stored: launch k reads its inputs on a 2-pixel ring and evaluates the stencil twice (Ap_{k-1} again on the 1-ring, Ap_k on its own pixels);
//   * Gauss-Newton keeps no residual vector either: p_{k-1} = M r_{k-1} + beta_{k-2} p_{k-2} determines r_{k-1}, so the state is a ring of three p buffers
//     (IterK::rfree); the first two launches of a solve read the solver's true r_0;
//   * Mask / Constraints are one flag byte, cos / sin come from the 4-byte angle, on a unit lattice U is folded into the arithmetic and the Jacobi
//     preconditioner is a 15-entry table indexed by the flag byte (PRE == 3); any other UrShape streams U and rebuilds M_a from the pairs it evaluates anyway (PRE == 2);
//   * delta is touched every second launch (two terms at once; p_{k-2} is kept in registers from its load).
// Structure: a workgroup owns a column strip and a contiguous range of rows; a lane keeps three rows of p_{k-1} and three of p_k of its column in registers
// (trip y turns the freshly loaded row y+2 into Ap_{k-1}(y+1), r_k, p_k(y+1), then Ap_k(y)); horizontal neighbours are whole-wave DPP shifts (a wave covers
// 64 pixels and produces the inner 60); three raw row buffers are requested three rows ahead and rotated by name; rows are addressed through buffer
// descriptors (soffset = row, voffset = lane constant: no address VALU); successive launches sweep top-down / bottom-up (FLIP) so that a launch starts on the
// rows the previous one left in the caches; MODE compiles the launch-to-launch state of the Gauss-Newton steady state in (1: odd launch, 2: even launch).
// LM = true: A = J^T J + diag(CtC), the restart launch after a split residual reset (energy.h PcgIterArgs), and
//   * general UrShape (PRE == 1): r, CtC, the preconditioner and b in memory, Q = 1/2 sum delta . (r + b) summed where delta is updated (solver.t:483-485);
//   * unit lattice (PRE == 3, round 6): no residual vector either -- the ring of three p buffers as in Gauss-Newton (1 / M is the table's own denominator), true r only in
//     the two launches behind PCGInit1 or a reset -- and no b: one CG step changes Q(delta) = b . delta - 1/2 delta^T A delta by alpha (p . r) - 1/2 alpha^2 (p . A p), and
//     every launch sums p_k . r_k beside its four other sums (in exact arithmetic it equals alphaNum_k; summed directly it also holds behind a reset, whose fresh r is not
//     orthogonal to the old p), so Q_k = Q_{k-1} + alpha_k (p_k . r_k - 1/2 alpha_k alphaDen_k) in the prologue of the launch that applies alpha_k.
//     Thread 0 of workgroup 0 keeps the running Q (IterK::qState) and publishes it where the host's early-out test polls (one tagged word pair instead of one per
//     workgroup, and at the START of the launch).  A reset re-anchors it to the host's direct sum (IterK::qInit).  With Q off delta's back, delta is paired as in
//     Gauss-Newton (written by every second launch into the other buffer, so an early-out still finds the old one; the term a deferring launch owes is added by
//     the solver before a reset, an early-out or the end of the loop: EnergyOps::iterFlushDelta).  89 -> 53 B/pixel.
#pragma once
#include "iw_device.h"

namespace optamd {
namespace {

constexpr int kSpan2 = kWave - 4;      // pixels a wave produces per row (two DPP rings)
// Workgroup size: 768 threads (3 waves per SIMD, 114-137 VGPRs) for the float unit-lattice kernel and, since its addresses moved to buffer descriptors, the float
// general Gauss-Newton kernel; its LM variant (200+ VGPRs) 512; double 256 (measured: DESIGN.md section 3).
template <class T, bool LATTICE, int PRE, bool LM> struct IterBlk {
    static constexpr int value = sizeof(T) == 8 ? 256 : LATTICE ? 768 : (PRE == 2 && !LM) ? 768 : 512;
};

template <class T>
struct IterRaw {           // one pixel's loads, untouched (any ALU op here would force a wait before the loop back-edge)
    V2<T> ro, po, mo, u, co; T ra, pa, ma, ca, ang;      // r (or p_{k-2}), p, M, UrShape, CtC; Offset part / Angle part; the angle
    int f, ok;
};
template <class T>
struct IterK {             // kernel argument block
    const T *rOld, *pOld; T *rNew, *pNew; T* delta; T* deltaOut;      // deltaOut == delta: in place
    const T* pre;          // the solver's 3-channel Jacobi preconditioner (PRE == 1)
    int first;             // first launch of a linear solve: alpha = beta = 0, r as given
    // deltaMode 0: delta += alpha_{k-1} p_{k-1} in every launch (LM with a general UrShape).  Paired (the r-free loops): 2 = this launch leaves delta alone, 1 = this launch applies the two
    // pending terms alpha_{k-2} p_{k-2} + alpha_{k-1} p_{k-1} in the reference's order -- 24 B/px every second launch instead of every launch.
    int deltaMode; const T* alphaIn; T* alphaOut;      // alpha_{k-2}, beta_{k-2} written by the previous launch ([0], [2]) / where this launch leaves its own
    int rfree;             // 0: r in memory (LM with a general UrShape);  1: rOld holds p_{k-2}, r rebuilt;  2: the two launches behind PCGInit1 / a reset -- rOld holds the solver's true r, r is not written either
    const T* CtC; const T* b; double* q; unsigned qTag; int afterReset; const double* betaNum; int nBetaNum; const double* betaDen; int nBetaDen;      // LM
    double* qState; double qInit;      // LM on a unit lattice (r-free): the running Q of the recurrence (one device double) / its value after a split residual reset (the host's direct sum)
    const double* prPrev; int nPr; double* pr;      // ... and the partial sums of p . r the previous launch left / this launch leaves (device memory, the energy's own)
    T lmRadius, lmMin, lmMax;      // PRE == 3 with LM: CtC and the LM preconditioner are rebuilt from the flag byte
    const double *aNumPrev, *aDenPrev, *s2Prev, *s3Prev; int nNum, nDen, n2, n3;
    double *aNum, *aDen, *s2, *s3;
    int ownBegin, ownEnd;  // slab mode: the launch may also update ghost rows (A.yBegin / A.yEnd include them); sums and delta stay on the owned image rows
    MailRefDev mail;       // slab mode, posted all-reduce: where the prologue polls the previous launch's four sums (words == nullptr: they are in aNumPrev .. s3Prev)
    MailPostDev post;      // ... and where this launch's last workgroup posts its own (world == 0: it does not)
    int deltaZero;         // the delta buffer has not been written since PCGInit1 and stands for 0 (honoured by the MODE 0 kernels)
};
template <class T>
struct IterBufs {          // descriptors of the arrays touched row by row, and the per-lane parts of the offsets
    __amdgpu_buffer_rsrc_t rOld, pOld, pNew, rNew, delta, deltaOut, angle, flags, pre, ctc, b, ur;
    unsigned x2, x1, x0;   // x * sizeof(V2<T>), x * sizeof(T), x  (x clamped into the row)
    unsigned aPart;        // 2 * N * sizeof(T): where the Angle part of a solver vector starts
};
template <class T, bool LATTICE, int PRE, bool LMV, bool FLIP>
__device__ __forceinline__ IterRaw<T> iw_iterLoad(const IWArgs<T>& A, const IterBufs<T>& B, bool xok, int y) {
    IterRaw<T> r;
    r.ok = xok && y >= 0 && y < A.H;
    const int yc = min(max(y, 0), A.H - 1);      // clamped: always a valid address, gated by r.ok
    const unsigned row = (unsigned)(FLIP ? A.H - 1 - yc : yc) * (unsigned)A.W;      // wave-uniform
    const unsigned s2 = row * (unsigned)sizeof(V2<T>), s1 = row * (unsigned)sizeof(T), s1a = s1 + B.aPart;
    const T* tag = nullptr;
    r.f = __builtin_amdgcn_raw_buffer_load_b8(B.flags, (int)B.x0, (int)row, 0);
    r.ro = bufLd2(B.rOld, B.x2, s2, tag); r.ra = bufLd1(B.rOld, B.x1, s1a, tag);
    r.po = bufLd2(B.pOld, B.x2, s2, tag); r.pa = bufLd1(B.pOld, B.x1, s1a, tag);
    if (PRE == 1) { r.mo = bufLd2(B.pre, B.x2, s2, tag); r.ma = bufLd1(B.pre, B.x1, s1a, tag); }
    else { r.mo = V2<T>{0, 0}; r.ma = 0; }
    if (LMV) { r.co = bufLd2(B.ctc, B.x2, s2, tag); r.ca = bufLd1(B.ctc, B.x1, s1a, tag); } else { r.co = V2<T>{0, 0}; r.ca = 0; }
    r.ang = bufLd1(B.angle, B.x1, s1, tag);
    if (LATTICE) r.u = V2<T>{0, 0}; else r.u = bufLd2(B.ur, B.x2, s2, tag);
    return r;
}

template <bool RIGHT, bool LATTICE, class T> __device__ __forceinline__ void dppShiftConst(const Q<T>& p, Q<T>& q) {   // the fields that do not change between p_{k-1} and p_k
    q.c = dppShift<RIGHT>(p.c); q.s = dppShift<RIGHT>(p.s); q.on = dppShift<RIGHT>(p.on);
    if (LATTICE) { q.ux = 0; q.uy = 0; } else { q.ux = dppShift<RIGHT>(p.ux); q.uy = dppShift<RIGHT>(p.uy); }
    q.fw = 0;
}
template <bool RIGHT, class T> __device__ __forceinline__ void dppShiftVec(const Q<T>& p, Q<T>& q) {
    q.ox = dppShift<RIGHT>(p.ox); q.oy = dppShift<RIGHT>(p.oy); q.a = dppShift<RIGHT>(p.a);
}
template <class T>
struct OldRow {            // one row of iteration k-1: p_{k-1} and, while still needed, r_{k-1}, M and (LM) CtC
    Q<T> q;
    T rx, ry, ra, mx, my, ma;
    T cx, cy, ca;
    T p2x, p2y, p2a;       // r-free loop: p_{k-2} of this pixel as loaded (the deferred delta term of an even launch needs it exactly)
};
template <class T>
struct NewRow {            // one row of iteration k: p_k, r_k, M, and the shifted constant fields of its neighbours
    Q<T> q;
    T rx, ry, ra, mx, my, ma;      // (z_k = M r_k is consumed where it is formed; the sums use M, r, A p themselves in double)
    T cx, cy, ca;          // CtC (LM)
    Q<T> lf, rt;           // only c, s, (ux, uy,) on are kept here
};

template <class T, bool LATTICE, int PRE, bool FLIP, bool LM = false, int MODE = 0>
__global__ __launch_bounds__((IterBlk<T, LATTICE, PRE, LM>::value), 1) void iw_pcgIter2(IWArgs<T> A, IterK<T> K, int rowsPerGroup, int gx, int gy) {
    static_assert(MODE == 0 || !LM || PRE == 3, "steady-state specialisations: the r-free loops (Gauss-Newton; Levenberg-Marquardt on a unit lattice)");
    static_assert(PRE >= 1 && PRE <= 3, "image_warping always preconditions (image_warping.t:10)");
    const int kDeltaMode = MODE == 1 ? 2 : MODE == 2 ? 1 : K.deltaMode, kRfree = MODE ? 1 : K.rfree;
    constexpr bool LMRF = LM && PRE == 3;      // Levenberg-Marquardt without r and b in memory (see the header)
    constexpr int kBlk = IterBlk<T, LATTICE, PRE, LM>::value, kStripW = (kBlk / kWave) * kSpan2;
    __shared__ double scratch[5 * (kBlk / kWave + 1)];
    const long N = (long)A.W * A.H;
    const int bx = blockIdx.x % gx, by = blockIdx.x / gx;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int x = bx * kStripW + wave * kSpan2 + lane - 2;
    const bool xok = x >= 0 && x < A.W;
    const bool writer = xok && lane >= 2 && lane < 2 + kSpan2;
    const int lyBegin = FLIP ? A.H - A.yEnd : A.yBegin, lyEnd = FLIP ? A.H - A.yBegin : A.yEnd;     // the rows this launch updates, in sweep coordinates
    const int yb = lyBegin + by * rowsPerGroup, ye = min(yb + rowsPerGroup, lyEnd);
    // The first five rows are requested before anything else: they do not depend on the scalars of the previous launch, so their latency
    // overlaps the prologue's own memory round trip (the partial sums another kernel just wrote) instead of following it (-1.5 us per launch).
    IterBufs<T> Bf;
    {
        const unsigned xc = (unsigned)min(max(x, 0), A.W - 1);
        Bf.x2 = xc * (unsigned)sizeof(V2<T>); Bf.x1 = xc * (unsigned)sizeof(T); Bf.x0 = xc; Bf.aPart = (unsigned)(2 * N * (long)sizeof(T));
        Bf.rOld = iw_rsrc(K.rOld); Bf.pOld = iw_rsrc(K.pOld); Bf.pNew = iw_rsrc(K.pNew); Bf.rNew = iw_rsrc(K.rNew); Bf.delta = iw_rsrc(K.delta); Bf.deltaOut = iw_rsrc(K.deltaOut);
        Bf.angle = iw_rsrc(A.Angle); Bf.flags = iw_rsrc(A.flags); Bf.pre = iw_rsrc(K.pre); Bf.ctc = iw_rsrc(K.CtC); Bf.b = iw_rsrc(K.b); Bf.ur = iw_rsrc(A.UrShape);
    }
    auto loadRow = [&](int y) { return iw_iterLoad<T, LATTICE, PRE, LM && PRE != 3, FLIP>(A, Bf, xok, y); };
    const IterRaw<T> raw0 = loadRow(yb - 2), raw1 = loadRow(yb - 1);
    IterRaw<T> rwA = loadRow(yb), rwB = loadRow(yb + 1), rwC = loadRow(yb + 2);
    T alpha = 0, beta = 0;
    const bool first = MODE ? false : K.first != 0;
    const bool restart = LM && MODE == 0 && K.afterReset != 0;      // r and delta are already those of this iteration (split residual reset)
    if (restart) {
        const double* const ps[2] = {K.betaNum, K.betaDen}; const int ns[2] = {K.nBetaNum, K.nBetaDen}; double o2[2];
        sumPartialsN<2>(ps, ns, scratch, o2);
        const T bNum = (T)o2[0], bDen = (T)o2[1];
        beta = (bDen > T(0)) ? bNum / bDen : T(0);     // solver.t:544-547
    } else if (!first) {
        const double* const ps[4] = {K.aNumPrev, K.aDenPrev, K.s2Prev, K.s3Prev}; const int ns[4] = {K.nNum, K.nDen, K.n2, K.n3}; double o4[4]; double prD = 0;
        if (!LM && K.mail.words) {                     // slab mode: the sums were posted to this rank's mailbox by every rank and may still be in flight
            __shared__ double mailScr[4 + 1 + 64];
            pollMailSums<4>(K.mail, mailScr, o4);
        } else if constexpr (LMRF) {                   // ... and p . r of the previous launch for the Q recurrence
            const double* const ps5[5] = {K.aNumPrev, K.aDenPrev, K.s2Prev, K.s3Prev, K.prPrev}; const int ns5[5] = {K.nNum, K.nDen, K.n2, K.n3, K.nPr}; double o5[5];
            sumPartialsN<5>(ps5, ns5, scratch, o5);
            o4[0] = o5[0]; o4[1] = o5[1]; o4[2] = o5[2]; o4[3] = o5[3]; prD = o5[4];
        } else sumPartialsN<4>(ps, ns, scratch, o4);   // the four sums of the previous launch, loads in flight together
        const double aNumD = o4[0], aDenD = o4[1], s2 = o4[2], s3 = o4[3];
        const T aNum = (T)aNumD, aDen = (T)aDenD;
        alpha = (aDen > T(0)) ? aNum / aDen : T(0);
        if (LMRF && blockIdx.x == 0 && threadIdx.x == 0) {      // Q of the iteration this launch applies, by the recurrence (header); the host polls word pair 0
            const double qNow = K.qState[0] + (double)alpha * (prD - 0.5 * (double)alpha * aDenD);
            K.qState[0] = qNow;
            if (K.qTag) storeTaggedPartial(K.q, 0, qNow, K.qTag); else K.q[0] = qNow;
        }
        // betaNumerator = sum M r_k^2 by expansion (energy.h); the reference's direct sum cannot be negative, so cancellation
        // noise below zero (residual dropping by >~1e3 in one iteration) is clamped away
        const double bNumD = fmax(aNumD - 2.0 * (double)alpha * s2 + (double)alpha * (double)alpha * s3, 0.0);
        beta = (aNum > T(0)) ? (T)bNumD / aNum : T(0);
    }
    if (K.alphaOut && blockIdx.x == 0 && threadIdx.x == 0) { K.alphaOut[0] = alpha; K.alphaOut[2] = beta; }
    if (LMRF && (first || restart) && blockIdx.x == 0 && threadIdx.x == 0) K.qState[0] = first ? 0.0 : K.qInit;      // Q_0 = 0 (delta = 0) / the direct sum of the reset
    const T alpha2 = (kDeltaMode == 1) ? K.alphaIn[0] : T(0);
    const bool reconR = (!LM || LMRF) && kRfree == 1;
    const T betaOlder = reconR ? K.alphaIn[2] : T(0);      // the beta of the previous launch: p_{k-1} = M r_{k-1} + betaOlder p_{k-2}
    auto phys = [&](int y) { return FLIP ? A.H - 1 - y : y; };   // sweep row -> image row
    const T w2 = A.w_reg * A.w_reg, wf2 = A.w_fit * A.w_fit;
    double accDen = 0, accNum = 0, acc2 = 0, acc3 = 0, accQ = 0;
    const bool keepR = first || restart;
    // M = guardedInvert(diag J^T J) of the Offset part takes one of 10 values whatever UrShape is (2 w^2 per active neighbour + w_fit^2), of the Angle part on a unit
    // lattice one of 5: a table indexed by the fit bit and the neighbour count of the flag byte.  The Offset entries repeat iw_evalJTF's accumulation order, so
    // they are the values the solver's preconditioner vector holds, bit for bit; the Angle entries use |R'(a) n|^2 = 1 exactly where iw_evalJTF rounds cos^2 + sin^2.
    __shared__ T mTab[16], cTab[16], iTab[16];      // iTab = 1 / M = (1 + sqrt(d))^2 directly (r-free loop)
    if (PRE == 3 || PRE == 2) {
        if (threadIdx.x < 15) {
            const int t = threadIdx.x, cnt = t < 10 ? t % 5 : t - 10;
            const T w = A.w_reg;
            T d = 0;
            if (t < 10) { for (int n = 0; n < cnt; ++n) d += w * w + w * w; if (t >= 5) d += A.w_fit * A.w_fit; }
            else for (int n = 0; n < cnt; ++n) d += (w * T(1)) * (w * T(1));
            const T sq = T(1) + sqrt(d);
            const T gi = T(1) / (sq * sq);                   // guardedInvert (solver.t:323-332)
            if (LM) {   // k_finalizeDiagonal (solver.t:631-664) on the table: SSq is the first outer iteration's guardedInvert(diag), and diag does not change
                const T radius = K.lmRadius, unclamped = d * (T(1) / radius), clampMul = (T(1) / gi) / radius;
                const T c = fmin(fmax(unclamped, K.lmMin * clampMul), K.lmMax * clampMul);
                cTab[t] = c; mTab[t] = T(1) / (c + radius * unclamped); iTab[t] = c + radius * unclamped;
            } else { mTab[t] = gi; iTab[t] = sq * sq; }
        }
        __syncthreads();
    }

    auto makeOld = [&](const IterRaw<T>& wRaw, OldRow<T>& o) {
        // The fields that enter the row window unchanged go through a real register move: otherwise the window field IS the load's destination register, the next
        // request for the buffer needs another one, and the compiler restores the names with copies at the back-edge -- copies of registers whose loads were issued
        // a moment ago: `s_waitcnt vmcnt(0)` once per pass, the whole prefetch drained every third row.
        IterRaw<T> w = wRaw;
        w.po.x = regCopy(wRaw.po.x); w.po.y = regCopy(wRaw.po.y); w.pa = regCopy(wRaw.pa);
        w.ro.x = regCopy(wRaw.ro.x); w.ro.y = regCopy(wRaw.ro.y); w.ra = regCopy(wRaw.ra);
        w.f = regCopy(wRaw.f);
        o.q.ox = w.po.x; o.q.oy = w.po.y; o.q.a = w.pa;
        { T sn, cn; sincosT(w.ang, &sn, &cn); o.q.c = cn; o.q.s = sn; }      // the same sincos as iw_cossin: the values a table would hold
        if (LATTICE) { o.q.ux = 0; o.q.uy = 0; } else { o.q.ux = w.u.x; o.q.uy = w.u.y; }
        o.q.on = (w.ok && (w.f & kActive)) ? T(1) : T(0);
        o.q.fw = (w.f & kFit) ? wf2 : T(0);
        o.rx = w.ro.x; o.ry = w.ro.y; o.ra = w.ra;
        if (!(LM && PRE == 3)) { o.cx = w.co.x; o.cy = w.co.y; o.ca = w.ca; }
        T ix = 1, iy = 1, ia = 1;
        const int cnt = (w.f >> kCountShift) & 7, io = cnt + ((w.f & kFit) ? 5 : 0);
        if (PRE == 3) {
            o.mx = o.my = mTab[io]; o.ma = mTab[10 + cnt];
            if (LM) { o.cx = o.cy = cTab[io]; o.ca = cTab[10 + cnt]; }
            if (reconR) { ix = iy = iTab[io]; ia = iTab[10 + cnt]; }
        } else if (PRE == 2) {      // M_a follows from the pairs of the row's first stencil evaluation (trip): until then the Angle part of a rebuilt r stays unscaled
            o.mx = o.my = mTab[io]; o.ma = 0;
            if (!LM && reconR) { ix = iy = iTab[io]; ia = T(1); }
        } else {
            o.mx = w.mo.x; o.my = w.mo.y; o.ma = w.ma;
            if (!LM && reconR) { ix = T(1) / o.mx; iy = T(1) / o.my; ia = T(1) / o.ma; }
        }
        if ((!LM || LMRF) && LATTICE && kRfree) { o.p2x = w.ro.x; o.p2y = w.ro.y; o.p2a = w.ra; }      // (the general-UrShape kernel has no registers to spare: it reads p_{k-2} again)
        if (reconR) {      // r_{k-1} = (p_{k-1} - beta p_{k-2}) / M: w.ro / w.ra were loaded from the p_{k-2} buffer
            o.rx = (o.q.ox - betaOlder * w.ro.x) * ix; o.ry = (o.q.oy - betaOlder * w.ro.y) * iy; o.ra = (o.q.a - betaOlder * w.ra) * ia;
        }
    };
    // J^T J at centre c; prev / next are the rows before / after it in sweep order.  Each pair of residuals is formed ONCE (iw_device.h): the right-hand pair of
    // lane x is the left-hand pair of lane x + 1 (three DPP moves), the pair towards the next row of the march is the pair towards the previous row one trip later.
    // `vert`: in, what the previous trip's evaluation of this stream left for the pair (prev, c); out, the same for (c, next)
    // wantM (PRE == 2, the p_{k-1} stream): pa receives diag(J^T J) of the Angle unknown -- the four pairs' terms in iw_evalJTF's order (right, left, image row y+1, y-1), the
    // left and the inherited vertical one taken from the neighbour's side of the pair (vertM: in, the previous trip's; out, this trip's)
    auto applyA = [&](const Q<T>& c, const Q<T>& lf, const Q<T>& rt, const Q<T>& prev, const Q<T>& next, PairOut<T>& vert, T& ox, T& oy, T& oa, bool wantM, T& vertM, T& pa) {
        T ax = 0, ay = 0, aa = 0;
        T mh[2] = {0, 0}, mv[2] = {0, 0};
        const PairOut<T> hr = iw_pairFull<1, 0, LATTICE>(c, rt, ax, ay, aa, wantM ? mh : nullptr, A.w_reg);
        PairOut<T> hl; hl.dx = dppShift<true>(hr.dx); hl.dy = dppShift<true>(hr.dy); hl.tn = dppShift<true>(hr.tn);
        const T mhl = wantM ? dppShift<true>(mh[1]) : T(0);
        iw_pairInherited(hl, lf.on, ax, ay, aa);
        if (!FLIP) {
            const PairOut<T> vn = iw_pairFull<0, 1, LATTICE>(c, next, ax, ay, aa, wantM ? mv : nullptr, A.w_reg); iw_pairInherited(vert, prev.on, ax, ay, aa); vert = vn;
            if (wantM) { T t = rt.on * mh[0]; t += lf.on * mhl; t += next.on * mv[0]; t += prev.on * vertM; pa = t; vertM = mv[1]; }
        } else {      // (image rows y+1, y-1 in that order in both directions)
            iw_pairInherited(vert, prev.on, ax, ay, aa); vert = iw_pairFull<0, -1, LATTICE>(c, next, ax, ay, aa, wantM ? mv : nullptr, A.w_reg);
            if (wantM) { T t = rt.on * mh[0]; t += lf.on * mhl; t += prev.on * vertM; t += next.on * mv[0]; pa = t; vertM = mv[1]; }
        }
        ox = c.on * (w2 * ax + c.fw * c.ox); oy = c.on * (w2 * ay + c.fw * c.oy); oa = c.on * (w2 * aa);
    };
    PairOut<T> vOld{0, 0, 0}, vNew{0, 0, 0};      // the vertical pairs the two stencil evaluations of a trip inherit (p_{k-1} rows / p_k rows)
    T vOldM = 0, unusedM = 0;                     // ... and, for M_a, the neighbour-side term of the p_{k-1} stream's vertical pair
    // The delta of the row a trip updates (y + 1) is requested one trip ahead, before that trip's prefetch of a raw row: by the time it is used a whole trip
    // has passed and the wait leaves the younger requests in flight, where a request at the point of use is the newest one and its wait (vmcnt(0)) drains
    // the whole queue once per row.  Every launch of the general-UrShape LM loop and the even launches of the r-free steady states (MODE 2) update delta in every
    // trip and take this form; the others read it where they use it.
    constexpr bool kDeltaEarly = MODE == 2 || (LM && !LMRF);
    struct DeltaPre { V2<T> o; T a; };
    auto loadDelta = [&](int y1) {
        DeltaPre d{V2<T>{0, 0}, 0};
        if constexpr (kDeltaEarly) {
            const int yc = min(max(y1, 0), A.H - 1);
            const unsigned rowE = (unsigned)(FLIP ? A.H - 1 - yc : yc) * (unsigned)A.W;
            const T* const tag = nullptr;
            d.o = bufLd2(Bf.delta, Bf.x2, rowE * (unsigned)sizeof(V2<T>), tag); d.a = bufLd1(Bf.delta, Bf.x1, rowE * (unsigned)sizeof(T) + Bf.aPart, tag);
            __builtin_amdgcn_sched_barrier(0);      // a side effect as far as code motion is concerned: the two requests stay here instead of being sunk into the branch that uses them
        }
        return d;
    };
    // One trip: the freshly loaded row y+2 -> Ap_{k-1}(y+1), r_k, z_k, p_k (y+1) -> Ap_k(y).
    // oA, oB = p_{k-1} rows y, y+1 (oC receives y+2);  nA, nB = p_k rows y-1, y (nC receives y+1)
    auto trip = [&](int y, const OldRow<T>& oA, const OldRow<T>& oB, const OldRow<T>& oC,
                    const NewRow<T>& nA, const NewRow<T>& nB, NewRow<T>& nC, bool live, const DeltaPre& dPre) {
        nC.q = oB.q;
        nC.lf = Q<T>{}; nC.lf.on = dppShift<true>(oB.q.on);      // of the left neighbour only its activity is needed: its pair comes ready-made
        dppShiftConst<false, LATTICE>(oB.q, nC.rt);
        Q<T> lf = nC.lf, rt = nC.rt;
        dppShiftVec<false>(oB.q, rt);
        T ax, ay, aa, paB = 0;
        applyA(oB.q, lf, rt, oA.q, oC.q, vOld, ax, ay, aa, PRE == 2, vOldM, paB);       // Step1 of iteration k-1 again
        if (LM) { ax += oB.cx * oB.q.ox; ay += oB.cy * oB.q.oy; aa += oB.ca * oB.q.a; }                                                       // + CtC p (o.t:2076-2082)
        T maB = oB.ma, raB = oB.ra;
        if (PRE == 2) {      // guardedInvert(diag J^T J) of the Angle unknown (solverGPUGaussNewton.t:323-332; iw_jtfMarch's mA)
            const T sq = T(1) + sqrt(paB), ia = sq * sq;
            maB = T(1) / ia;
            if (!LM && reconR) raB = oB.ra * ia;      // the rebuilt residual's Angle part, scaled now that 1 / M_a is known
        }
        const T rx = keepR ? oB.rx : oB.rx - alpha * ax, ry = keepR ? oB.ry : oB.ry - alpha * ay, ra = keepR ? raB : raB - alpha * aa;   // Step2
        nC.mx = oB.mx; nC.my = oB.my; nC.ma = maB;
        nC.cx = oB.cx; nC.cy = oB.cy; nC.ca = oB.ca;
        nC.rx = rx; nC.ry = ry; nC.ra = ra;
        const T zx = nC.mx * rx, zy = nC.my * ry, za = nC.ma * ra;
        nC.q.ox = zx + beta * oB.q.ox; nC.q.oy = zy + beta * oB.q.oy; nC.q.a = za + beta * oB.q.a;                                        // Step3
        if (live && writer && y + 1 >= yb && y + 1 < ye) {
            const int yp = phys(y + 1);
            const unsigned rowE = (unsigned)yp * (unsigned)A.W, s2 = rowE * (unsigned)sizeof(V2<T>), s1a = rowE * (unsigned)sizeof(T) + Bf.aPart;      // wave-uniform row offsets
            const T* const tag = nullptr;
            const bool own = yp >= K.ownBegin && yp < K.ownEnd;
            if (own && !keepR && kDeltaMode != 2) {   // delta += alpha_{k-1} p_{k-1}  (solver.t:461-462), preceded by the deferred term of launch k-1
                V2<T> d; T da;
                if (kDeltaEarly) { d = dPre.o; da = dPre.a; }
                else { d = bufLd2(Bf.delta, Bf.x2, s2, tag); da = bufLd1(Bf.delta, Bf.x1, s1a, tag); }
                if (MODE == 0 && K.deltaZero) { d.x = 0; d.y = 0; da = 0; }      // first delta update of a linear solve whose PCGInit1 left the buffer untouched
                if (kDeltaMode == 1) {
                    if ((!LM || LMRF) && LATTICE && kRfree == 1) { d.x += alpha2 * oB.p2x; d.y += alpha2 * oB.p2y; da += alpha2 * oB.p2a; }      // p_{k-2} kept from the load: exact
                    else {      // p_{k-2} from memory: the buffer read through rOld (r-free ring), or the p buffer about to be overwritten
                        const __amdgpu_buffer_rsrc_t qb = ((!LM || LMRF) && kRfree) ? Bf.rOld : Bf.pNew;
                        const V2<T> q = bufLd2(qb, Bf.x2, s2, tag); const T qa = bufLd1(qb, Bf.x1, s1a, tag);
                        d.x += alpha2 * q.x; d.y += alpha2 * q.y; da += alpha2 * qa;
                    }
                }
                d.x += alpha * oB.q.ox; d.y += alpha * oB.q.oy; da += alpha * oB.q.a;
                bufSt2(Bf.deltaOut, Bf.x2, s2, d.x, d.y); bufSt1(Bf.deltaOut, Bf.x1, s1a, da);
                if (LM && !LMRF) {   // Q = 1/2 sum delta . (r + b) with the updated delta and r (solver.t:483-485)
                    const V2<T> bo = bufLd2(Bf.b, Bf.x2, s2, tag); const T ba = bufLd1(Bf.b, Bf.x1, s1a, tag);
                    accQ += (double)(T(0.5) * (d.x * (rx + bo.x))) + (double)(T(0.5) * (d.y * (ry + bo.y))) + (double)(T(0.5) * (da * (ra + ba)));
                }
            }
            if (!kRfree) { bufSt2(Bf.rNew, Bf.x2, s2, rx, ry); bufSt1(Bf.rNew, Bf.x1, s1a, ra); }
            bufSt2(Bf.pNew, Bf.x2, s2, nC.q.ox, nC.q.oy); bufSt1(Bf.pNew, Bf.x1, s1a, nC.q.a);
        }
        Q<T> l2 = nB.lf, r2 = nB.rt;
        dppShiftVec<false>(nB.q, r2);
        T ox, oy, oa;
        applyA(nB.q, l2, r2, nA.q, nC.q, vNew, ox, oy, oa, false, unusedM, unusedM);     // Step1 of iteration k
        if (LM) { ox += nB.cx * nB.q.ox; oy += nB.cy * nB.q.oy; oa += nB.ca * nB.q.a; }
        if (live && writer && y >= yb && phys(y) >= K.ownBegin && phys(y) < K.ownEnd) {
            accDen += (double)(nB.q.ox * ox + nB.q.oy * oy + nB.q.a * oa);
            if (LMRF) accQ += (double)nB.q.ox * (double)nB.rx + (double)nB.q.oy * (double)nB.ry + (double)nB.q.a * (double)nB.ra;      // p_k . r_k, exact products
            // sum M r^2, sum M r Ap, sum M Ap^2 of this row from shared double factors.  The expansion of the beta numerator cancels to as many digits as the residual
            // loses in one iteration, so every term is formed from the same M, r, A p in double, where a product of two floats is exact.
            const double mx = (double)nB.mx, my = (double)nB.my, ma = (double)nB.ma;
            const double rx2 = (double)nB.rx, ry2 = (double)nB.ry, ra2 = (double)nB.ra, ax2 = (double)ox, ay2 = (double)oy, az2 = (double)oa;
            const double mrx = mx * rx2, mry = my * ry2, mra = ma * ra2;
            accNum += mrx * rx2 + mry * ry2 + mra * ra2;
            acc2 += mrx * ax2 + mry * ay2 + mra * az2;
            acc3 += (mx * ax2) * ax2 + (my * ay2) * ay2 + (ma * az2) * az2;
        }
    };
    OldRow<T> o0, o1, o2;
    NewRow<T> n0{}, n1{}, n2{};
    makeOld(raw0, o0);
    makeOld(raw1, o1);
    { T t0 = 0, t1 = 0, t2 = 0, m0[2] = {0, 0}; vOld = iw_pairFull<0, FLIP ? -1 : 1, LATTICE>(o0.q, o1.q, t0, t1, t2, PRE == 2 ? m0 : nullptr, A.w_reg); vOldM = m0[1]; }      // the pair (row yb-2, row yb-1) the first trip inherits
    DeltaPre dlA = loadDelta(yb - 1), dlB = dlA, dlC = dlA;      // (trip yb - 2 updates no row; its delta is a dummy)
    // trips y = yb-2 .. ye-1 (the first two only build p_k(yb-1), p_k(yb)); three per pass, no branch around a load; the barrier keeps the strip's waves on the same rows
    for (int y = yb - 2; y < ye; y += 3) {
        __syncthreads();
        // a raw row is consumed into the row window before its buffer is requested again; the request still precedes the trip's arithmetic
        { makeOld(rwA, o2); dlB = loadDelta(y + 2); rwA = loadRow(y + 5); trip(y, o0, o1, o2, n0, n1, n2, true, dlA); }
        { makeOld(rwB, o0); dlC = loadDelta(y + 3); rwB = loadRow(y + 6); trip(y + 1, o1, o2, o0, n1, n2, n0, y + 1 < ye, dlB); }
        { makeOld(rwC, o1); dlA = loadDelta(y + 4); rwC = loadRow(y + 7); trip(y + 2, o2, o0, o1, n2, n0, n1, y + 2 < ye, dlC); }
    }
    double v[5] = {accDen, accNum, acc2, acc3, accQ};
    blockReduceSumN<5>(v, scratch);
    if (threadIdx.x == 0) {
        K.aDen[blockIdx.x] = v[0]; K.aNum[blockIdx.x] = v[1]; K.s2[blockIdx.x] = v[2]; K.s3[blockIdx.x] = v[3];
        if (LMRF) K.pr[blockIdx.x] = v[4];
        else if (LM) { if (K.qTag) storeTaggedPartial(K.q, blockIdx.x, v[4], K.qTag); else K.q[blockIdx.x] = v[4]; }
    }
    if (!LM && K.post.world) {      // slab mode: the last workgroup to finish posts the four sums (order of the consumer's poll: aNum, aDen, s2, s3) to every rank's mailbox
        double* const parts[4] = {K.aNum, K.aDen, K.s2, K.s3};
        postMailSums<4>(K.post, parts, scratch);
    }
}

}  // namespace
}  // namespace optamd
