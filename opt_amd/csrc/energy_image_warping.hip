// image_warping: 2-D as-rigid-as-possible image warp -- the north-star workload.
//
// Energy (reference examples/image_warping/image_warping.t:1-23): per pixel c, unknowns O_c (float2
// offset) and a_c (angle); for the 4 stencil directions n
//     r_reg[c,n] = w_reg * v(c,n) * [ (O_c - O_{c+n}) - R(a_c)(U_c - U_{c+n}) ],  v = InBounds(c+n) & Mask_{c+n}=0 & Mask_c=0
//     r_fit[c]   = w_fit * [C_c >= 0] * (O_c - C_c)
// cost = 1/2 sum r^2 over non-masked pixels (Exclude, :11).  The kernels are the hand-derived
// counterparts of what Opt's generator emits from that file (o.t:2029-2089 applyJTJ, :2129-2172 evalJTF,
// :2375-2385 cost, :2174-2225 modelcost); with D_{c,n} = R'(a_c)(U_c - U_{c+n}):
//     J p  at (c,n)      : Jp = w [ (pO_c - pO_{c+n}) - D_{c,n} pa_c ]
//     (J^T J p)_O(c)     = w_fit^2 f_c pO_c + w sum_n v [ Jp(c,n) - Jp(c+n,-n) ]
//     (J^T J p)_a(c)     = - w sum_n v D_{c,n} . Jp(c,n)
//
// Where the kernels live (all HBM-bound streaming / stencil work, ~3 flop/B: no MFMA anywhere):
//   iw_device.h   pixel record, the residual-pair arithmetic, buffer-descriptor addressing
//   iw_step.h     what runs once per Gauss-Newton step (flags + lattice verdict, PCGInit1, cost, update) and iw_applyJTJ (probes, LM reset, three-kernel loop)
//   iw_iter.h     iw_pcgIter2: one launch per PCG iteration, 53 B/pixel (the reference's three kernels: 180) -- images that do not fit the chip, row slabs, LM
//   iw_onchip.h   iw_onchipPcg: the whole linear solve as one persistent launch with its state in registers / LDS -- images of up to 2 M pixels
// This file is the host side: which kernel runs when.
#include "energy.h"
#include "iw_device.h"
#include "iw_step.h"
#include "iw_iter.h"
#include "iw_onchip.h"
#include <cstdint>

namespace optamd {
namespace {

template <class T>
struct ImageWarpingOps : EnergyOps<T> {
    IWArgs<T> A{};
    int cus = 256;
    // Switches (environment, read once per plan).  OPT_AMD_LATTICE=0: take the general-UrShape kernels whatever the input (bench.py's general_urshape leg, tests);
    // OPT_AMD_ITER_ROWS=R: every row-marching workgroup takes R rows (tests run small images in the benchmark's regime); OPT_AMD_ITER_MAXWG=n: no row-marching
    // launch uses more than n workgroups (ranks that SHARE one GPU: see maxWorkgroups); OPT_AMD_SLAB_PERIOD=1: exchange ghost rows after every launch;
    // OPT_AMD_ONCHIP*: the on-chip linear solve (below).  The reference-ordered three-kernel loop is the solver's OPT_AMD_ONEKERNEL=0.
    bool useLattice = true;
    int forceRows = 0, maxWorkgroups = 1 << 30, maxExchangePeriod = 1 << 20;
    ImageWarpingOps(const unsigned* dims) {
        A.W = (int)dims[0]; A.H = (int)dims[1];
        this->usePreconditioner = true;                                           // image_warping.t:10
        this->addUnknown(0, (long)A.W * A.H, 2); this->addUnknown(1, (long)A.W * A.H, 1);   // Offset, Angle (:2-3)
        HIP_CHECK(hipMalloc((void**)&A.flags, (size_t)A.W * A.H));
        HIP_CHECK(hipMalloc((void**)&A.cs, (size_t)A.W * A.H * 2 * sizeof(T)));
        int dev = 0; HIP_CHECK(hipGetDevice(&dev));
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (const char* e = getenv("OPT_AMD_LATTICE")) useLattice = atoi(e) != 0;
        if (const char* e = getenv("OPT_AMD_SLAB_PERIOD")) maxExchangePeriod = std::max(1, atoi(e));
        if (const char* e = getenv("OPT_AMD_ITER_ROWS")) forceRows = std::max(0, atoi(e));
        if (const char* e = getenv("OPT_AMD_ITER_MAXWG")) maxWorkgroups = std::max(1, atoi(e));
        if (const char* e = getenv("OPT_AMD_ONCHIP")) ocEnabled = atoi(e) != 0;
        if (const char* e = getenv("OPT_AMD_ONCHIP_ROWS")) ocForceRows = std::max(0, atoi(e));
        if (const char* e = getenv("OPT_AMD_ONCHIP_FLAT")) ocFlatMax = std::max(0, atoi(e));
        if (const char* e = getenv("OPT_AMD_ONCHIP_FAIL_AT")) ocFailAt = atoi(e);      // test hook: see OnchipArgs::failAt
        if (const char* e = getenv("OPT_AMD_ONCHIP_FAIL_LAUNCH")) ocFailLaunch = atoi(e);      // ... in the n-th on-chip launch of the plan only (default: in every one)
        if (const char* e = getenv("OPT_AMD_ONCHIP_TIMEOUT_MS")) ocTimeoutTicks = std::max(1, atoi(e)) * 100000LL;
        HIP_CHECK(hipMalloc((void**)&dNotLattice, sizeof(int)));
        HIP_CHECK(hipHostMalloc((void**)&hNotLattice, 64)); *hNotLattice = 0;
        HIP_CHECK(hipEventCreateWithFlags(&bindEvent, hipEventDisableTiming));
    }
    ~ImageWarpingOps() override {
        if (ocS.slots) { (void)hipFree(ocS.slots); (void)hipFree(ocS.groupSlots); (void)hipFree(ocS.inbox); }
        if (ocS.bad) { (void)hipFree(ocS.bad); (void)hipHostFree(ocS.hostErr); }
        (void)hipHostFree(hNotLattice); (void)hipEventDestroy(bindEvent);
        for (T* b : ring) if (b) (void)hipFree(b);
        (void)hipFree(A.flags); (void)hipFree(A.cs); if (alphaSlots) (void)hipFree(alphaSlots); (void)hipFree(dNotLattice);
        if (lmQState) (void)hipFree(lmQState);
    }
    int flatGrid(long n) const { return (int)std::max<long>(1, std::min<long>((n + kBlock - 1) / kBlock, std::min<long>(kMaxPartials, (long)cus * 8))); }

    // ---- bind: flag bytes + is UrShape the unit lattice? (the reference example always passes the pixel grid, CombinedSolver.h:161-172) ---------------------
    int* hNotLattice = nullptr; int* dNotLattice = nullptr; hipEvent_t bindEvent = nullptr; bool verdictPending = false, lattice = false;
    bool bindInvariantDuringSolve() const override { return true; }      // the flag bytes and the lattice verdict depend on Mask, Constraints and UrShape only
    void bind(void** p, LaunchCtx& ctx) override {
        A.Offset = (const T*)p[0]; A.Angle = (const T*)p[1]; A.UrShape = (const T*)p[2]; A.Constraints = (const T*)p[3]; A.Mask = (const T*)p[4];
        A.w_fit = (T) * (const float*)p[5]; A.w_reg = (T) * (const float*)p[6];   // Param(..., float, ...) stays float in double mode (:7-8)
        const Slab& s = this->slab;
        if (s.active) { A.yBegin = s.yBegin; A.yEnd = s.yEnd; A.gy0 = s.gy0; A.Hg = s.Hg; }
        else { A.yBegin = 0; A.yEnd = A.H; A.gy0 = 0; A.Hg = A.H; }
        if (!s.active) {
            // one marching pass: flag bytes + the unit-lattice verdict, which lands in pinned memory and is read when it is first needed (resolveLattice) --
            // nothing blocks here.  Until then `lattice` keeps the previous bind's verdict as a hint (false before the first).
            ScopedKernel k(ctx, "bindFlags");
            if (verdictPending) resolveLattice();      // (two binds without a consumer in between: the older verdict is not needed any more, but its event is)
            *hNotLattice = 0;
            int gx, gy, rpg; marchGrid(A.yEnd - A.yBegin, gx, gy, rpg);
            if (useLattice) iw_bindMarch<T, true><<<gx * gy, kBlock, 0, ctx.stream>>>(A, hNotLattice, rpg, gx, gy);
            else iw_bindMarch<T, false><<<gx * gy, kBlock, 0, ctx.stream>>>(A, hNotLattice, rpg, gx, gy);
            HIP_CHECK(hipEventRecord(bindEvent, ctx.stream));
            verdictPending = useLattice;
            if (!useLattice) lattice = false;
            return;
        }
        // row slabs: the one-thread-per-pixel kernels (ghost rows beyond the global image count as non-existent), verdict read back at once
        { ScopedKernel k(ctx, "bindFlags"); iw_flags<T><<<flatGrid((long)A.W * A.H), kBlock, 0, ctx.stream>>>(A); }
        lattice = false; verdictPending = false;
        if (useLattice) {
            ScopedKernel k(ctx, "checkLattice");
            int h = 1;
            HIP_CHECK(hipMemsetAsync(dNotLattice, 0, sizeof(int), ctx.stream));
            iw_checkLattice<T><<<flatGrid((long)A.W * A.H), kBlock, 0, ctx.stream>>>(A, dNotLattice);
            HIP_CHECK(hipMemcpyAsync(&h, dNotLattice, sizeof(int), hipMemcpyDeviceToHost, ctx.stream));
            HIP_CHECK(hipStreamSynchronize(ctx.stream));
            lattice = (h == 0);
        }
    }
    // The verdict of the last iw_bindMarch: waits for that kernel only (an event), not for what was enqueued behind it.
    bool resolveLattice() {
        if (verdictPending) {
            HIP_CHECK(hipEventSynchronize(bindEvent));
            lattice = __atomic_load_n(hNotLattice, __ATOMIC_ACQUIRE) == 0;
            verdictPending = false;
        }
        return lattice;
    }
    T* unknownPtr(int img) const override { return const_cast<T*>(img == 0 ? A.Offset : A.Angle); }

    // ---- grids: one co-resident wave of workgroups, rows split evenly ---------------------------------------------------------------------------------------
    int occMarch = 0;
    void marchGrid(int rows, int& gx, int& gy, int& rowsPerGroup) {
        if (occMarch == 0) {      // 256-thread workgroups, ~60 VGPRs: the co-resident count of the widest of the marching kernels
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occMarch, (const void*)iw_jtfMarch<T, false, false>, kBlock, 0));
            occMarch = std::max(1, std::min(occMarch, 8));
        }
        gx = divUp(A.W, kStrip);
        splitRows(rows, gx, cus * occMarch, gy, rowsPerGroup);
    }
    // maxWorkgroups: an iteration kernel that polls a posted all-reduce in its prologue must be co-resident with the peers' kernels it is waiting for, which holds
    // on one GPU per rank and on a shared GPU only while all ranks' workgroups together fit the chip (bench.py --share-gpu, tests/test_peer_comm_gpu.py).
    void splitRows(int rows, int gx, int target, int& gy, int& rowsPerGroup) const {
        target = std::max(gx, std::min(target, maxWorkgroups));
        gy = std::max(1, std::min(std::min(rows, target / gx), kMaxPartials / gx));
        rowsPerGroup = divUp(rows, gy);
        if (forceRows > 0) rowsPerGroup = std::max(divUp(rows, std::max(1, kMaxPartials / gx)), std::min(rows, forceRows));
        gy = divUp(rows, rowsPerGroup);
    }

    // ---- once per Gauss-Newton step ----------------------------------------------------------------------------------------------------------------------------
    bool fastGN() const {      // single GPU, vectors below 4 GiB (buffer-descriptor offsets are 32-bit): the marching PCGInit1 and the single-kernel / on-chip loops
        return !this->slab.active && (unsigned long long)A.W * A.H * 3ull * sizeof(T) < (1ull << 32);
    }
    T *initR = nullptr, *initP = nullptr; Reduction* initRed = nullptr; bool initHint = false, initPending = false, deltaZero = false;
    void launchJtf(bool lat, LaunchCtx& ctx, Reduction* cost = nullptr, const JtfLm<T>* lmInit = nullptr) {
        ScopedKernel k(ctx, cost ? "computeCost+PCGInit1" : "PCGInit1");
        int gx, gy, rpg; marchGrid(A.yEnd - A.yBegin, gx, gy, rpg);
        if (cost) {
            if (lat) iw_jtfMarch<T, true, true><<<gx * gy, kBlock, 0, ctx.stream>>>(A, initR, initP, initRed->partials, cost->partials, rpg, gx, gy);
            else iw_jtfMarch<T, false, true><<<gx * gy, kBlock, 0, ctx.stream>>>(A, initR, initP, initRed->partials, cost->partials, rpg, gx, gy);
            cost->n = gx * gy;
        } else if (lmInit) {
            if (lat) iw_jtfMarch<T, true, false, true><<<gx * gy, kBlock, 0, ctx.stream>>>(A, initR, initP, initRed->partials, nullptr, rpg, gx, gy, *lmInit);
            else iw_jtfMarch<T, false, false, true><<<gx * gy, kBlock, 0, ctx.stream>>>(A, initR, initP, initRed->partials, nullptr, rpg, gx, gy, *lmInit);
        } else if (lat) iw_jtfMarch<T, true, false><<<gx * gy, kBlock, 0, ctx.stream>>>(A, initR, initP, initRed->partials, nullptr, rpg, gx, gy);
        else iw_jtfMarch<T, false, false><<<gx * gy, kBlock, 0, ctx.stream>>>(A, initR, initP, initRed->partials, nullptr, rpg, gx, gy);
        initRed->n = gx * gy;
    }
    // PCGInit1 + PCGInit1_Finish for the Gauss-Newton loops: r = -J^T F, p = M r, partial sums of r.p -- one marching kernel (no cos/sin table, no diag /
    // preconditioner vectors: the loops take M from the flag byte and, for a general UrShape, from the pairs they evaluate).  delta = 0 (solver.t:389) is not written: the first launch that updates delta takes
    // it as 0 (IterK::deltaZero), finishUpdate / pcgFinish do the same if no launch did.  The kernel variant follows the lattice verdict of the previous bind
    // while this bind's is still in flight; the loops check it before their first launch.
    bool evalJTFInit(T* r, T* p, T* /*delta*/, long /*nPad*/, Reduction& aNum0, LaunchCtx& ctx) override {
        if (!fastGN()) return false;
        initR = r; initP = p; initRed = &aNum0; initHint = lattice;
        launchJtf(initHint, ctx);
        deltaZero = true; initPending = true;
        return true;
    }
    // The cost of the step that has just finished and PCGInit1 of the next one, one march (inside Opt_ProblemSolve: solver.hip).  The lattice verdict is resolved here -- the
    // cost needs it anyway -- so the variant launched is final and nothing is left pending.
    bool evalCostAndJTFInit(Reduction& cost, T* r, T* p, T* /*delta*/, long /*nPad*/, Reduction& aNum0, LaunchCtx& ctx) override {
        if (!fastGN()) return false;
        initR = r; initP = p; initRed = &aNum0; initHint = resolveLattice();
        launchJtf(initHint, ctx, &cost);
        deltaZero = true; initPending = true;
        return true;
    }
    // Levenberg-Marquardt's PCGInit1 + PCGSaveSSq + PCGFinalizeDiagonal as one march (iw_jtfMarch<.., LMINIT>; single GPU).  The cost of this Init / Step has been read by
    // now, so the lattice verdict is known and the variant launched is final.
    bool evalJTFInitLM(const LmInitArgs<T>& a, LaunchCtx& ctx) override {
        if (!fastGN()) return false;
        const bool lat = resolveLattice();
        initR = a.r; initP = a.p; initRed = a.rDotP; initHint = lat; initPending = false;
        const JtfLm<T> L{a.CtC, a.SSq, a.delta, a.pre, a.b, a.radius, a.minLm, a.maxLm, a.saveSSq, a.q->partials};
        launchJtf(lat, ctx, nullptr, &L);
        a.q->n = a.rDotP->n;
        deltaZero = false;      // (this pass writes delta = 0 itself: the LM loops read it)
        return true;
    }
    void evalCost(Reduction& out, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "computeCost");
        if (!this->slab.active) {
            const bool lat = resolveLattice();
            int gx, gy, rpg; marchGrid(A.yEnd - A.yBegin, gx, gy, rpg);
            if (lat) iw_costMarch<T, true><<<gx * gy, kBlock, 0, ctx.stream>>>(A, out.partials, rpg, gx, gy);
            else iw_costMarch<T, false><<<gx * gy, kBlock, 0, ctx.stream>>>(A, out.partials, rpg, gx, gy);
            out.n = gx * gy;
            return;
        }
        const int g = flatGrid((long)A.W * (A.yEnd - A.yBegin));
        iw_cost<T><<<g, kBlock, 0, ctx.stream>>>(A, out.partials);
        out.n = g;
    }
    void evalJTF(T* r, T* diag, LaunchCtx& ctx) override {      // the generic form: probes, LM, row slabs
        const int g = flatGrid((long)A.W * A.H);
        { ScopedKernel k(ctx, "cosSinTable"); iw_cossin<T><<<g, kBlock, 0, ctx.stream>>>(A); }
        { ScopedKernel k(ctx, "PCGInit1"); iw_evalJTF<T><<<g, kBlock, 0, ctx.stream>>>(A, r, diag); }
    }
    void evalModelCost(const T* delta, Reduction& out, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "computeModelCost");
        const int g = flatGrid((long)A.W * (A.yEnd - A.yBegin));
        iw_modelCost<T><<<g, kBlock, 0, ctx.stream>>>(A, delta, out.partials);
        out.n = g;
    }

    // ---- J^T J p as its own kernel (probes, the LM residual reset, the three-kernel loop) ------------------------------------------------------------------------
    int occ[2][2] = {{0, 0}, {0, 0}};
    int blocksPerCU(bool lmv, bool fused) {
        int& o = occ[lmv][fused];
        if (o == 0) {
            const void* fn = lmv ? (fused ? (const void*)iw_applyJTJ<T, true, true> : (const void*)iw_applyJTJ<T, true, false>)
                                 : (fused ? (const void*)iw_applyJTJ<T, false, true> : (const void*)iw_applyJTJ<T, false, false>);
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, fn, kBlock, 0));
            o = std::max(1, std::min(o, 8));
        }
        return o;
    }
    void launchApply(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx, const FuseArgs<T>* fuse) {
        const int gx = divUp(A.W, kStrip);
        int gy, rowsPerGroup;
        splitRows(A.yEnd - A.yBegin, gx, cus * blocksPerCU(CtC != nullptr, fuse != nullptr), gy, rowsPerGroup);
        const int grid = gx * gy;
        {
            ScopedKernel k(ctx, fuse ? "PCGStep3+PCGStep1" : "PCGStep1");
            double* part = dot ? dot->partials : nullptr;
            FuseArgs<T> F = fuse ? *fuse : FuseArgs<T>{};
            if (fuse) {
                if (CtC) iw_applyJTJ<T, true, true><<<grid, kBlock, 0, ctx.stream>>>(A, v, out, CtC, part, rowsPerGroup, gx, gy, F);
                else iw_applyJTJ<T, false, true><<<grid, kBlock, 0, ctx.stream>>>(A, v, out, nullptr, part, rowsPerGroup, gx, gy, F);
            } else {
                if (CtC) iw_applyJTJ<T, true, false><<<grid, kBlock, 0, ctx.stream>>>(A, v, out, CtC, part, rowsPerGroup, gx, gy, F);
                else iw_applyJTJ<T, false, false><<<grid, kBlock, 0, ctx.stream>>>(A, v, out, nullptr, part, rowsPerGroup, gx, gy, F);
            }
            if (dot) dot->n = grid;
        }
        if (this->slab.active) iw_zeroGhost<T><<<divUp(A.W, kBlock), kBlock, 0, ctx.stream>>>(A, out);
    }
    void applyJTJ(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx) override { launchApply(v, out, CtC, dot, ctx, nullptr); }
    // computeAdelta + PCGStep2_2ndHalf of LM's split residual reset (solver.t:1079-1083) in one march: A delta is consumed where it is formed (single GPU)
    int occReset = 0;
    bool applyJTJResetLM(const T* delta, T* r, const T* b, const T* pre, T* z, const T* CtC, Reduction& bNum, Reduction& q, LaunchCtx& ctx) override {
        if (this->slab.active || !pre || !CtC) return false;
        if (occReset == 0) {
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occReset, (const void*)iw_applyJTJ<T, true, false, true>, kBlock, 0));
            occReset = std::max(1, std::min(occReset, 8));
        }
        const int gx = divUp(A.W, kStrip);
        int gy, rowsPerGroup;
        splitRows(A.yEnd - A.yBegin, gx, cus * occReset, gy, rowsPerGroup);
        FuseArgs<T> F{};
        F.resetB = b; F.resetPre = pre; F.resetR = r; F.resetZ = z; F.resetBNum = bNum.partials; F.resetQ = q.partials;
        ScopedKernel k(ctx, "PCGStep1+PCGStep2_2ndHalf");
        iw_applyJTJ<T, true, false, true><<<gx * gy, kBlock, 0, ctx.stream>>>(A, delta, nullptr, CtC, nullptr, rowsPerGroup, gx, gy, F);
        bNum.n = gx * gy; q.n = gx * gy;
        return true;
    }
    bool applyJTJFused(const T* pOld, const T* z, T* pNew, T* out, const T* CtC, Reduction* dot, const Reduction& bNum, const double* aNumOld,
                       double* aNumNext, LaunchCtx& ctx) override {
        FuseArgs<T> F{z, pNew, bNum.partials, bNum.n, aNumOld, aNumNext};
        launchApply(pOld, out, CtC, dot, ctx, &F);
        return true;
    }

    // ---- one launch per PCG iteration (iw_iter.h) ---------------------------------------------------------------------------------------------------------------------
    static const void* iterKernel(bool lat, bool lmLoop, bool flip, int mode) {      // mode: 0 launch state from the arguments, 1 / 2 steady state of the Gauss-Newton loop (odd / even launch)
#define IWK(LAT, PRE, LMV, MODE) (flip ? (const void*)iw_pcgIter2<T, LAT, PRE, true, LMV, MODE> : (const void*)iw_pcgIter2<T, LAT, PRE, false, LMV, MODE>)
        if (lmLoop) return lat ? (mode == 1 ? IWK(true, 3, true, 1) : mode == 2 ? IWK(true, 3, true, 2) : IWK(true, 3, true, 0)) : IWK(false, 1, true, 0);
        if (lat) return mode == 1 ? IWK(true, 3, false, 1) : mode == 2 ? IWK(true, 3, false, 2) : IWK(true, 3, false, 0);
        return mode == 1 ? IWK(false, 2, false, 1) : mode == 2 ? IWK(false, 2, false, 2) : IWK(false, 2, false, 0);
#undef IWK
    }
    static int iterBlock(bool lat, bool lmLoop) { return sizeof(T) == 8 ? 256 : lat ? 768 : lmLoop ? 512 : 768; }      // IterBlk<T, LATTICE, PRE, LM>::value of iterKernel's choice
    int occIter[4] = {0, 0, 0, 0};
    int iterFlip = 0, sinceExchange = 0, iterIndex = 0;
    bool deferredTerm = false, lastLoopRfree = false, lastLoopLmRing = false, lastWroteDelta = true; int sinceTrueR = 0, lmPrN = 0; double* lmQState = nullptr;
    const T* owedP[2] = {nullptr, nullptr};
    T* ring[3] = {nullptr, nullptr, nullptr}; const T* r0Ptr = nullptr; T* alphaSlots = nullptr;
    // What the loops do before their first launch: this bind's lattice verdict (the marching bind does not block for it); a PCGInit1 that ran on the previous
    // verdict and guessed "lattice" for an input that is none is redone.
    void beginLoop(LaunchCtx& ctx) {
        resolveLattice();
        if (initPending) { if (initHint && !lattice) launchJtf(false, ctx); initHint = lattice; initPending = false; }
    }
    bool pcgIteration(const PcgIterArgs<T>& a, LaunchCtx& ctx) override {
        // A p is recomputed, not stored, so a slab needs two ghost rows (one for each stencil evaluation); with one the solver runs the three-kernel loop.
        if (!a.pre || (this->slab.active && this->slab.ghost < 2)) return false;
        if ((unsigned long long)A.W * A.H * 3ull * sizeof(T) >= (1ull << 32)) return false;      // 32-bit buffer offsets
        const bool lmLoop = a.CtC != nullptr;
        if (lmLoop && this->slab.active) return false;      // the LM variant is single-GPU
        if (a.first) beginLoop(ctx);
        this->iterStateExchange = true;      // slab mode: the ghost rows of the loop state come from the neighbours after a launch
        this->iterTakesMail = !lmLoop;       // the prologue can poll a posted all-reduce
        // With g >= 2 ghost rows whose state is current to depth v, a launch can also update the ghost rows to depth v - 1 (their A p needs one more row on either
        // side) and its sums need depth 2; so after an exchange at depth g the slab runs g - 1 launches before it needs the neighbours again, launch j = 1 .. g - 1
        // of the period updating g - j ghost rows (none in the last one: they are about to be overwritten).
        int ext = 0;
        this->iterExchangeDue = true;
        if (this->slab.active) {
            if (a.first) sinceExchange = 0;
            const int period = std::max(1, std::min(this->slab.ghost - 1, maxExchangePeriod)), j = sinceExchange + 1;
            const bool due = j >= period;
            ext = due ? 0 : this->slab.ghost - j;
            sinceExchange = due ? 0 : j;
            this->iterExchangeDue = due;
        }
        IWArgs<T> Ax = A;                   // what the kernel sees: the rows it updates
        Ax.yBegin = std::max(0, A.yBegin - ext); Ax.yEnd = std::min(A.H, A.yEnd + ext);
        if (a.first) { iterFlip = 0; iterIndex = 0; }      // every linear solve starts top-down, so a solve is reproducible whatever ran before it
        const int blk = iterBlock(lattice, lmLoop), L = (lmLoop ? 2 : 0) + (lattice ? 1 : 0);
        if (occIter[L] == 0) {
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occIter[L], iterKernel(lattice, lmLoop, false, 0), blk, 0));
            occIter[L] = std::max(1, std::min(occIter[L], 8));
        }
        const int gx = divUp(A.W, (blk / kWave) * kSpan2);
        int gy, rowsPerGroup;
        splitRows(Ax.yEnd - Ax.yBegin, gx, cus * occIter[L], gy, rowsPerGroup);
        // Gauss-Newton: delta every second launch, and no residual vector -- the state is a ring of three p buffers (ring[j % 3] holds p_j; the first two
        // launches read the solver's r_0).  LM on a unit lattice (round 6) runs on the same ring -- the two launches behind PCGInit1 or a split residual reset read the true r the
        // solver holds -- and takes Q from the CG recurrence (iw_iter.h), so neither r nor b moves; delta is updated in every launch (an early-out must find it complete).
        // LM with a general UrShape keeps r, CtC, M and b in memory.
        const bool gn = !lmLoop, lmRing = lmLoop && lattice;
        const T *rOldPtr = a.rOld, *pOldPtr = a.pOld; T* pNewPtr = a.pNew; int rfreeFlag = 0, deltaMode = 0; const T* alphaIn = nullptr; T* alphaOut = nullptr;
        if (gn || lmRing) {
            const size_t bytes = ((size_t)A.W * A.H * 3 + 3) / 4 * 4 * sizeof(T);      // padded like the solver's vectors: its flat kernels read whole 16-byte packs of the last p
            for (int j = 0; j < 3; ++j) if (!ring[j]) { HIP_CHECK(hipMalloc((void**)&ring[j], bytes)); HIP_CHECK(hipMemsetAsync(ring[j], 0, bytes, ctx.stream)); }
            if (a.first || a.afterReset) { r0Ptr = a.rOld; sinceTrueR = 0; }      // the solver swaps its r buffers after every launch; this one keeps the true r until the second launch has read it
            const int k = iterIndex;
            pOldPtr = k == 0 ? a.pOld : ring[(k - 1) % 3];
            rOldPtr = sinceTrueR <= 1 ? r0Ptr : ring[(k - 2) % 3];
            pNewPtr = ring[k % 3];
            rfreeFlag = sinceTrueR <= 1 ? 2 : 1;
            if (!alphaSlots) { HIP_CHECK(hipMalloc((void**)&alphaSlots, 4 * sizeof(T))); HIP_CHECK(hipMemsetAsync(alphaSlots, 0, 4 * sizeof(T), ctx.stream)); }   // [0,1] alpha, [2,3] beta, ping-pong
            deltaMode = (sinceTrueR >= 2 && sinceTrueR % 2 == 0) ? 1 : 2;      // the launch behind PCGInit1 / a reset has nothing to apply; odd launches defer
            // what this launch leaves owed to delta (LM: the solver has it added before a reset, an early-out or the end of the loop -- iterFlushDelta)
            owedP[k & 1] = (lmRing && sinceTrueR % 2 == 1) ? pOldPtr : nullptr; lastWroteDelta = !lmRing || deltaMode == 1;
            ++sinceTrueR;
            alphaOut = alphaSlots + (k & 1); alphaIn = alphaSlots + ((k & 1) ^ 1);
        }
        if (lmRing && !lmQState) {      // the running Q and the two sets of p . r partials (this launch's / the previous launch's)
            HIP_CHECK(hipMalloc((void**)&lmQState, sizeof(double) * (1 + 2 * (size_t)kMaxPartials)));
            HIP_CHECK(hipMemsetAsync(lmQState, 0, sizeof(double) * (1 + 2 * (size_t)kMaxPartials), ctx.stream));
        }
        if (!gn && !lmRing) lastWroteDelta = true;
        lastLoopRfree = gn; lastLoopLmRing = lmRing;
        deferredTerm = gn && iterIndex >= 1 && iterIndex % 2 == 1;            // after an odd launch alpha_{k-1} p_{k-1} is still owed (pcgFinish / finishUpdate)
        IterK<T> K{};
        K.rOld = rOldPtr; K.pOld = pOldPtr; K.rNew = a.rNew; K.pNew = pNewPtr; K.delta = a.delta; K.deltaOut = a.deltaOut ? a.deltaOut : a.delta;
        K.pre = a.pre; K.first = a.first; K.deltaMode = deltaMode; K.alphaIn = alphaIn; K.alphaOut = alphaOut; K.rfree = rfreeFlag;
        K.CtC = a.CtC; K.b = a.b; K.q = a.q ? a.q->partials : nullptr; K.qTag = a.qTag; K.afterReset = a.afterReset;
        K.betaNum = a.betaNum.partials; K.nBetaNum = a.betaNum.n; K.betaDen = a.betaDen.partials; K.nBetaDen = a.betaDen.n;
        if (lmRing) {
            K.qState = lmQState; K.qInit = a.qInit;
            K.pr = lmQState + 1 + (size_t)(iterIndex & 1) * kMaxPartials; K.prPrev = lmQState + 1 + (size_t)((iterIndex & 1) ^ 1) * kMaxPartials; K.nPr = lmPrN;
        }
        K.lmRadius = a.lmRadius; K.lmMin = a.lmMinDiag; K.lmMax = a.lmMaxDiag;
        K.aNumPrev = a.aNumPrev.partials; K.aDenPrev = a.aDenPrev.partials; K.s2Prev = a.s2Prev.partials; K.s3Prev = a.s3Prev.partials;
        K.nNum = a.aNumPrev.n; K.nDen = a.aDenPrev.n; K.n2 = a.s2Prev.n; K.n3 = a.s3Prev.n;
        K.aNum = a.aNum->partials; K.aDen = a.aDen->partials; K.s2 = a.s2->partials; K.s3 = a.s3->partials;
        K.ownBegin = A.yBegin; K.ownEnd = A.yEnd;
        K.mail = MailRefDev{a.mail.words, a.mail.world, a.mail.stride, a.mail.tag, a.mail.timeoutTicks, a.mail.errFlag};
        for (int t = 0; t < 16; ++t) K.post.dst[t] = a.post.dst[t];
        K.post.world = a.post.world; K.post.tag = a.post.tag; K.post.ticket = a.post.ticket;
        K.deltaZero = 0;
        if (deltaZero && !a.first && deltaMode != 2 && gn) { K.deltaZero = 1; deltaZero = false; }      // this launch writes every pixel's delta: from here on the buffer is real
        {
            ScopedKernel k(ctx, "PCGIteration");
            int rpg = rowsPerGroup, gxa = gx, gya = gy;
            void* kargs[] = {(void*)&Ax, (void*)&K, (void*)&rpg, (void*)&gxa, (void*)&gya};
            // from the third launch of a Gauss-Newton solve on the launch state alternates between two values: compiled in
            const int mode = ((gn || lmRing) && rfreeFlag == 1 && !a.first && !a.afterReset && !K.deltaZero) ? (deltaMode == 2 ? 1 : 2) : 0;
            HIP_CHECK(hipLaunchKernel(iterKernel(lattice, lmLoop, iterFlip != 0, mode), dim3(gx * gy), dim3(blk), kargs, 0, ctx.stream));
        }
        iterFlip ^= 1;      // successive launches sweep top-down / bottom-up: a launch starts on the rows the previous one left in the caches
        ++iterIndex;
        a.aNum->n = a.aDen->n = a.s2->n = a.s3->n = gx * gy;
        if (a.q) a.q->n = lmRing ? 1 : gx * gy;      // (the recurrence's Q is one value, published by workgroup 0)
        lmPrN = gx * gy;
        return true;
    }
    // Where p of the last launch lives when the loop keeps its own buffers (the LM loop's reset and tail kernels read it)
    const T* iterCurrentP() const override { return lastLoopLmRing && iterIndex >= 1 ? ring[(iterIndex - 1) % 3] : nullptr; }
    bool iterWroteDelta() const override { return lastWroteDelta; }
    // delta += the term the launch `issuedBeyond` before the one issued last left owed (a deferring launch of the paired LM loop), if any
    bool iterOwedTerm(int issuedBeyond, const T** p, const T** alpha) const override {
        const int idx = iterIndex - 1 - issuedBeyond;
        if (!lastLoopLmRing || idx < 0 || !owedP[idx & 1]) return false;
        *p = owedP[idx & 1]; *alpha = alphaSlots + (idx & 1);
        return true;
    }
    void iterFlushDelta(T* delta, int issuedBeyond, LaunchCtx& ctx) override {
        const T *p = nullptr, *alpha = nullptr;
        if (!iterOwedTerm(issuedBeyond, &p, &alpha)) return;
        ScopedKernel k(ctx, "PCGStep2_delta");
        const long n = 3L * A.W * A.H;
        iw_axpyDeferred<T><<<flatGrid(n), kBlock, 0, ctx.stream>>>(delta, p, alpha, n);
    }
    // Slab mode, after launch iterIndex - 1: the vectors whose ghost rows the neighbours must refresh -- the two newest search directions of the ring.
    int iterExchangeVectors(T** out) override {
        if (!lastLoopRfree || iterIndex < 1) return 0;
        out[0] = ring[(iterIndex - 1) % 3];
        if (iterIndex < 2) return 1;                      // launch 1 reads the solver's r_0 (ghost rows exchanged before the loop) and p_0
        out[1] = ring[(iterIndex - 2) % 3];
        return 2;
    }
    // After the last launch L-1 of a linear solve.  If it was an odd launch, the term alpha_{L-2} p_{L-2} was deferred: its alpha sits in the slot that launch
    // wrote.  The solver then adds alpha_{L-1} p_{L-1}; returns where p_{L-1} lives.
    const T* pcgFinish(const T* pPrev, T* delta, LaunchCtx& ctx) override {
        if (deltaZero) { HIP_CHECK(hipMemsetAsync(delta, 0, ((size_t)A.W * A.H * 3 + 3) / 4 * 4 * sizeof(T), ctx.stream)); deltaZero = false; }      // the generic tail reads it
        const T* pLast = nullptr;
        if (lastLoopRfree && iterIndex >= 1) {
            pLast = ring[(iterIndex - 1) % 3];
            if (iterIndex >= 2) pPrev = ring[(iterIndex - 2) % 3];
        }
        if (!deferredTerm) return pLast;
        ScopedKernel k(ctx, "PCGStep2_delta");
        const long n = 3L * A.W * A.H;
        iw_axpyDeferred<T><<<flatGrid(n), kBlock, 0, ctx.stream>>>(delta, pPrev, alphaSlots + ((iterIndex - 1) & 1), n);
        deferredTerm = false;
        return pLast;
    }
    // Last delta terms + X += delta in one pass (iw_finishUpdate), single GPU.
    bool finishUpdate(const T* pPrev, const T* pLast, const T* delta, const Reduction& aNum, const Reduction& aDen, LaunchCtx& ctx) override {
        if (this->slab.active) return false;
        if (lastLoopRfree && iterIndex >= 1) {
            pLast = ring[(iterIndex - 1) % 3];
            if (iterIndex >= 2) pPrev = ring[(iterIndex - 2) % 3];
        }
        ScopedKernel k(ctx, "PCGLinearUpdate");
        const long N = (long)A.W * A.H;
        iw_finishUpdate<T><<<flatGrid(N), kBlock, 0, ctx.stream>>>(const_cast<T*>(A.Offset), const_cast<T*>(A.Angle), deltaZero ? nullptr : delta, pLast, deferredTerm ? pPrev : nullptr,
                                                                   alphaSlots ? alphaSlots + ((iterIndex - 1) & 1) : nullptr, N, aNum.partials, aNum.n, aDen.partials, aDen.n);
        deferredTerm = false; deltaZero = false;
        return true;
    }
    bool iterPostsItself(bool lmLoop) const override {      // would pcgIteration accept the launch (and so carry out a planned post)?  Same conditions as its refusals above.
        return !lmLoop && this->slab.active && this->slab.ghost >= 2 && (unsigned long long)A.W * A.H * 3ull * sizeof(T) < (1ull << 32);
    }
    bool slabIterationAvailable() const override { return this->slab.ghost >= 2 && (unsigned long long)A.W * A.H * 3ull * sizeof(T) < (1ull << 32); }
    bool supportsSlab() const override { return true; }
    long rowScalars(int img) const override { return (long)A.W * (img == 0 ? 2 : 1); }

    // ---- the whole linear solve on chip (iw_onchip.h): unit lattice, Gauss-Newton (also on row slabs) or Levenberg-Marquardt, tiles <= CUs ------------------------------------------
    // OPT_AMD_ONCHIP=0 switches it off (the one A/B switch of the path); OPT_AMD_ONCHIP_ROWS=r forces the variant with r rows per lane (tests run every
    // variant on small images); OPT_AMD_ONCHIP_FLAT=n: grids of up to n workgroups sum flat instead of through the two-level tree (same bits either way).
    struct OcVariant { int rows; bool apLds, deltaGlb; const void* fn; size_t lds; int occ; };
    std::vector<OcVariant> ocVariants, ocVariantsLM;
    bool ocEnabled = true, ocFailed = false, ocLaunched = false;
    int ocForceRows = 0, ocFlatMax = 256, ocFailAt = -1, ocFailLaunch = -1, ocLaunchCount = 0; long long* ocProf = nullptr; long long ocTimeoutTicks = 0;      // 0: onchip_sync.h ocTimeouts() decides; OPT_AMD_ONCHIP_TIMEOUT_MS overrides
    OnchipSync ocS{}; unsigned ocSeq = 1; size_t ocInboxBytes = 0, ocSlotBytes = 0, ocGroupBytes = 0;
    void ocInit() {
        if (!ocVariants.empty()) return;
        if constexpr (sizeof(T) == 4) {
            ocVariants.push_back({2, false, false, (const void*)iw_onchipPcg<T, 2, false, false>, OcLds<T>::total(2, false), 0});      // (small images: twice the tiles of ROWS = 4, half the serial work per lane: 512^2 5.5 -> 5.0 us per iteration)
            ocVariants.push_back({4, false, false, (const void*)iw_onchipPcg<T, 4, false, false>, OcLds<T>::total(4, false), 0});
            ocVariants.push_back({8, false, false, (const void*)iw_onchipPcg<T, 8, false, false>, OcLds<T>::total(8, false), 0});
            ocVariants.push_back({16, true, true, (const void*)iw_onchipPcg<T, 16, true, true>, OcLds<T>::total(16, true), 0});
            // Levenberg-Marquardt: A p and delta in registers, b in LDS (up to 4096 pixels per CU: 1 M pixels)
            ocVariantsLM.push_back({2, false, false, (const void*)iw_onchipPcg<T, 2, false, false, true>, OcLds<T>::total(2, false, true), 0});
            ocVariantsLM.push_back({4, false, false, (const void*)iw_onchipPcg<T, 4, false, false, true>, OcLds<T>::total(4, false, true), 0});
            ocVariantsLM.push_back({8, false, false, (const void*)iw_onchipPcg<T, 8, false, false, true>, OcLds<T>::total(8, false, true), 0});
        } else {
            ocVariants.push_back({2, false, false, (const void*)iw_onchipPcg<T, 2, false, false>, OcLds<T>::total(2, false), 0});
            ocVariants.push_back({4, false, false, (const void*)iw_onchipPcg<T, 4, false, false>, OcLds<T>::total(4, false), 0});      // (double: up to 2048 pixels per CU)
            ocVariantsLM.push_back({2, false, false, (const void*)iw_onchipPcg<T, 2, false, false, true>, OcLds<T>::total(2, false, true), 0});      // (double LM: b and the halo copies of delta do not fit the LDS at ROWS = 4)
        }
        for (auto* vs : {&ocVariants, &ocVariantsLM})
            for (auto& v : *vs) {
                if (hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds) != hipSuccess) { (void)hipGetLastError(); v.occ = 0; continue; }      // (a variant the device cannot hold is simply not offered)
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v.occ, v.fn, kOcBlock, v.lds) != hipSuccess) { (void)hipGetLastError(); v.occ = 0; }
                v.occ = std::min(v.occ, 1);      // one workgroup per CU: the co-residency the in-kernel waits rely on does not depend on how the dispatcher packs CUs
            }
    }
    // Which variant, if any: the smallest ROWS whose tiles fit one per CU.  A slab's owned rows must be whole tiles (its last tile row faces the next rank's first).
    const OcVariant* ocSelect(int& tX, int& tY, bool lmv = false) {
        ocInit();
        tX = divUp(A.W, kOcTileW);
        const int rowsOwned = A.yEnd - A.yBegin;
        for (const auto& v : lmv ? ocVariantsLM : ocVariants) {
            if (ocForceRows && v.rows != ocForceRows) continue;
            const int th = kOcWavesY * v.rows;
            if (this->slab.active && rowsOwned % th != 0) continue;
            tY = divUp(rowsOwned, th);
            if (v.occ >= 1 && (long)tX * tY <= std::min(std::min(cus * v.occ, kOcMaxTiles), maxWorkgroups)) return &v;
        }
        return nullptr;
    }
    bool slabOnChipAvailable(int L) override {      // (row slabs: the lattice verdict of this bind is already known, bind() read it back)
        int tX, tY;
        if (!(ocEnabled && !ocFailed && L > 0 && this->slab.active && this->slab.ghost >= 2 && this->onChipPlan && lattice &&
              (unsigned long long)A.W * A.H * 3ull * sizeof(T) < (1ull << 32) && ocSelect(tX, tY) != nullptr)) return false;
        // ... and the communicator could plan it (count = 0: a dry query -- its sticky error state and the capacity of its edge boxes -- so that a "no" is part of the vote)
        return this->onChipPlan(this->onChipCtx, 4, 0, tX, 3L * kOcTileW * (long)(sizeof(T) / 4), nullptr) != 0;
    }
    bool pcgSolveOnChip(const T* r0, const T* p0, T* delta, int L, double* traceDev, const OnChipLm<T>* lmArgs, LaunchCtx& ctx) override {
        const bool slabMode = this->slab.active;
        if (lmArgs && (slabMode || traceDev || lmArgs->resetPeriod < 1)) return false;      // the LM variants are single-GPU
        if (!ocEnabled || ocFailed || L <= 0 || (unsigned long long)A.W * A.H * 3ull * sizeof(T) >= (1ull << 32)) return false;
        if (slabMode && (traceDev || !slabOnChipAvailable(L))) return false;
        resolveLattice();
        if (initPending && initHint && !lattice) { launchJtf(false, ctx); initHint = false; }      // PCGInit1 ran on the previous bind's verdict (see beginLoop)
        if (!lattice) return false;
        int tX = 0, tY = 0;
        const OcVariant* V = ocSelect(tX, tY, lmArgs != nullptr);
        if (!V) return false;
        const int G = tX * tY;
        if (lmArgs && G > ocFlatMax) return false;      // (the LM variants sum flat)
        if (!ocS.slots) {      // sized for this plan's image once (the dimensions of a plan are fixed); zero = no tag
            const int maxRows = ocVariants.front().rows;
            const int gMax = std::min(kOcMaxTiles, tX * divUp(A.yEnd - A.yBegin, kOcWavesY * maxRows));
            ocS.stride = 3L * kOcTileW * (long)(sizeof(T) / 4);
            ocSlotBytes = sizeof(oc_u64) * 2 * (size_t)gMax * 2 * kOcSumsMax; ocGroupBytes = sizeof(oc_u64) * 2 * (size_t)divUp(gMax, kOcGroup) * 8;
            ocInboxBytes = sizeof(oc_u64) * 2 * (size_t)gMax * 4 * (size_t)ocS.stride;
            HIP_CHECK(hipMalloc((void**)&ocS.slots, ocSlotBytes)); HIP_CHECK(hipMalloc((void**)&ocS.groupSlots, ocGroupBytes)); HIP_CHECK(hipMalloc((void**)&ocS.inbox, ocInboxBytes));
            if (!ocS.bad) { HIP_CHECK(hipMalloc((void**)&ocS.bad, sizeof(int))); HIP_CHECK(hipHostMalloc((void**)&ocS.hostErr, 64)); for (int w_ = 0; w_ < 16; ++w_) ocS.hostErr[w_] = 0; HIP_CHECK(hipMemsetAsync(ocS.bad, 0, sizeof(int), ctx.stream)); }
            ocSeq = 0xE0000001u;      // forces the clearing below
#if OC_PROFILE
            if (getenv("OPT_AMD_ONCHIP_PROFILE")) { HIP_CHECK(hipMalloc((void**)&ocProf, sizeof(long long) * kOcWaves * 16 * kOcMaxTiles)); }
#endif
        }
        const unsigned nTags = lmArgs ? 2u * (unsigned)L : (unsigned)L;      // (an LM iteration that ends with the split residual reset has two phases)
        if (ocSeq > 0xE0000000u || ocSeq + nTags > 0xE0000000u) {      // tags never repeat: start over on cleared buffers long before the counter wraps
            HIP_CHECK(hipMemsetAsync(ocS.slots, 0, ocSlotBytes, ctx.stream)); HIP_CHECK(hipMemsetAsync(ocS.groupSlots, 0, ocGroupBytes, ctx.stream));
            HIP_CHECK(hipMemsetAsync(ocS.inbox, 0, ocInboxBytes, ctx.stream));
            ocSeq = 2;
        }
        OcLinks links{};
        if (slabMode) {      // (every rank makes this call: the solver has made the decision to run on chip collective)
            OptAmd_OnChipLinks Lk{};
            if (!this->onChipPlan(this->onChipCtx, 4, L, tX, ocS.stride, &Lk)) return false;
            for (int t = 0; t < 16; ++t) links.mailDst[t] = Lk.mailDst[t];
            links.mailMine = Lk.mailMine; links.world = Lk.world; links.rank = Lk.rank; links.slots = Lk.slots; links.slotStride = Lk.slotStride; links.rankStride = Lk.rankStride;
            links.seq0 = Lk.seq0; links.edgeSendUp = Lk.edgeSendUp; links.edgeSendDown = Lk.edgeSendDown; links.edgeRecvUp = Lk.edgeRecvUp; links.edgeRecvDown = Lk.edgeRecvDown;
            links.edgeParityStride = Lk.edgeParityStride;
        }
        const OcTimeouts tmo = ocTimeouts(ocTimeoutTicks, L, slabMode);
        OnchipArgs<T> K{A.W, A.H, tX, tY, G, A.yBegin, A.yEnd, links, r0, p0, A.Angle, A.flags, delta, A.w_fit, A.w_reg, L, ocSeq, G <= ocFlatMax ? 1 : 0, ocS, lmArgs ? lmArgs->breakInfo : traceDev,
                        tmo.later, tmo.first, ocProf, (ocFailLaunch < 0 || ocLaunchCount == ocFailLaunch) ? ocFailAt : -1, T(0), T(0), T(0), T(0), 1};
        ++ocLaunchCount;
        if (lmArgs) { K.lmRadius = lmArgs->radius; K.lmMin = lmArgs->minLm; K.lmMax = lmArgs->maxLm; K.qTolerance = lmArgs->qTolerance; K.resetPeriod = lmArgs->resetPeriod; }
        ocSeq += nTags;
        {
            ScopedKernel k(ctx, "PCGSolveOnChip");
            void* kargs[] = {(void*)&K};
            HIP_CHECK(hipLaunchKernel(V->fn, dim3(G), dim3(kOcBlock), kargs, V->lds, ctx.stream));
        }
        if (lmArgs) { iw_relayBad<<<1, kWave, 0, ctx.stream>>>(ocS.bad, ocS.hostErr); ocLaunched = true; }      // (the solver applies the update itself)
        else if (!slabMode) onChipApply(delta, nullptr, false, ctx);      // (row slabs: the solver all-reduces the ranks' verdicts first, then calls onChipApply)
#if OC_PROFILE
        if (ocProf) {      // development builds: where an iteration's time goes, per wave of a workgroup, mean over the workgroups
            std::vector<long long> h((size_t)G * kOcWaves * 16);
            HIP_CHECK(hipStreamSynchronize(ctx.stream));
            HIP_CHECK(hipMemcpy(h.data(), ocProf, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
            const char* names[9] = {"stencil", "send", "wave-sums", "barrier", "inbox", "grid-sum", "delta-req", "halo-upd", "update"};
            fprintf(stderr, "on-chip profile %dx%d rows=%d G=%d L=%d (us per iteration, mean over workgroups; one line per wave)\n   wave", A.W, A.H, V->rows, G, L);
            for (int ph = 0; ph < 9; ++ph) fprintf(stderr, " %9s", names[ph]);
            fprintf(stderr, "     total\n");
            for (int w = 0; w < kOcWaves; ++w) {
                fprintf(stderr, "   %4d", w);
                double tot = 0;
                for (int ph = 0; ph < 9; ++ph) {
                    double mean = 0;
                    for (int b = 0; b < G; ++b) mean += h[((size_t)b * kOcWaves + w) * 16 + ph] * 0.01 / L / G;
                    tot += mean;
                    fprintf(stderr, " %9.2f", mean);
                }
                fprintf(stderr, " %9.2f\n", tot);
            }
        }
#endif
        return true;
    }
    // PCGLinearUpdate behind the on-chip Gauss-Newton solve.  verdict (row slabs): device scalar, the number of ranks whose kernel failed -- all ranks apply or none.
    // refused: this rank launched no kernel at all (its peers will time out): its contribution to the verdict is "failed".
    void onChipVerdict(double* out, bool refused, LaunchCtx& ctx) override {
        if (!ocS.bad) { HIP_CHECK(hipMalloc((void**)&ocS.bad, sizeof(int))); HIP_CHECK(hipHostMalloc((void**)&ocS.hostErr, 64)); for (int w_ = 0; w_ < 16; ++w_) ocS.hostErr[w_] = 0; HIP_CHECK(hipMemsetAsync(ocS.bad, 0, sizeof(int), ctx.stream)); }
        iw_badToScalar<<<1, kWave, 0, ctx.stream>>>(ocS.bad, refused ? 1 : 0, out);
    }
    void onChipApply(const T* delta, const double* verdict, bool /*refused*/, LaunchCtx& ctx) override {
        ScopedKernel k(ctx, "PCGLinearUpdate");
        const long N = (long)A.W * A.H;
        iw_applyDelta<T><<<flatGrid(N), kBlock, 0, ctx.stream>>>(const_cast<T*>(A.Offset), const_cast<T*>(A.Angle), delta, N, ocS.bad, verdict, ocS.hostErr,
                                                                 ocStepSlot >= 0 ? ocS.hostErr + 1 + ocStepSlot : nullptr);
        ocLaunched = true;
    }
    std::string describe(int L, bool lmv) override {      // ("key=value; ..." -- no ';' inside a value)
        const Slab& sl = this->slab;
        const int rowsOwned = (sl.active ? sl.yEnd - sl.yBegin : A.H);
        char buf[900];
        if (sl.active) { A.yBegin = sl.yBegin; A.yEnd = sl.yEnd; } else { A.yBegin = 0; A.yEnd = A.H; }      // (what bind() will set: ocSelect reads them)
        int tX = 0, tY = 0;
        const bool eligible = ocEnabled && !ocFailed && L > 0 && !(lmv && sl.active);
        const OcVariant* V = eligible ? ocSelect(tX, tY, lmv) : nullptr;
        // the same question without the workgroup cap of ranks that SHARE a GPU (OPT_AMD_ITER_MAXWG): what one GPU per rank would answer
        int uX = 0, uY = 0; const OcVariant* U = nullptr;
        if (eligible && !V && maxWorkgroups < (1 << 30)) { const int keep = maxWorkgroups; maxWorkgroups = 1 << 30; U = ocSelect(uX, uY, lmv); maxWorkgroups = keep; }
        const bool ghostOk = !sl.active || sl.ghost >= 2;
        const long rowBytes = (long)A.W * 3 * (long)sizeof(T);
        const std::string cap = maxWorkgroups < (1 << 30) ? std::to_string(maxWorkgroups) + " (OPT_AMD_ITER_MAXWG: ranks share a GPU)" : std::string("none");
        if (V && ghostOk)
            snprintf(buf, sizeof buf, "path=on-chip (if UrShape is the unit lattice%s); onchip_rows_per_lane=%d; tiles=%dx%d of %d CUs; lds_bytes=%zu; slab_rows=%d; ghost_rows=%d; workgroup_cap=%s; "
                     "per_iteration_cross_rank=%s; fallback=one launch per PCG iteration (iw_pcgIter2)",
                     sl.active ? " and every rank and the communicator agree" : "", V->rows, tX, tY, cus, V->lds, rowsOwned, sl.active ? sl.ghost : 0, cap.c_str(),
                     sl.active ? "edge rows of A p as tagged words (2 x W x 3 scalars x 8 B per neighbour) + one rank hop of 4 doubles, inside the persistent launch" : "none");
        else {
            const std::string why = !ocEnabled ? "switched off" : ocFailed ? "a wait timed out earlier" : (lmv && sl.active) ? "the LM variants are single-GPU" : !ghostOk ? "needs >= 2 ghost rows" :
                                    U ? "the workgroup cap -- without it: on-chip, " + std::to_string(U->rows) + " rows per lane, " + std::to_string(uX) + "x" + std::to_string(uY) + " tiles" :
                                    "the tiles do not fit one per CU (or the slab is no whole number of tiles)";
            const std::string cross = sl.active ? "one all-reduce of 4 doubles, and every " + std::to_string(std::max(1, sl.ghost - 1)) + " iterations " + std::to_string(2L * sl.ghost * rowBytes) +
                                                  " B of edge rows (the two newest search directions) per neighbour" : std::string("none");
            snprintf(buf, sizeof buf, "path=one launch per PCG iteration (iw_pcgIter2%s); why_not_on_chip=%s; slab_rows=%d; ghost_rows=%d; workgroup_cap=%s; per_iteration_cross_rank=%s",
                     lmv ? ", LM" : "", why.c_str(), rowsOwned, sl.active ? sl.ghost : 0, cap.c_str(), cross.c_str());
        }
        return buf;
    }
    bool onChipFailed() override {
        if (!ocLaunched) return false;
        ocLaunched = false;
        if (__atomic_load_n(ocS.hostErr, __ATOMIC_ACQUIRE) == 0) return false;
        ocFailed = true;
        return true;
    }
    bool onChipFailedPeek() override { return ocLaunched && ocS.hostErr && __atomic_load_n(ocS.hostErr, __ATOMIC_ACQUIRE) != 0; }
    int ocStepSlot = -1;      // (hostErr is 16 ints: word 0 "some launch failed", words 1 .. 15 one per deferred step)
    bool supportsDeferredSteps() const override { return !this->slab.active; }
    bool deltaMovable() const override { return !this->slab.active; }
    void onChipStepSlot(int slot) override { ocStepSlot = (slot >= 0 && slot < 15) ? slot : -1; }
    bool onChipStepFailed(int slot) override { return ocS.hostErr && slot >= 0 && slot < 15 && __atomic_load_n(ocS.hostErr + 1 + slot, __ATOMIC_ACQUIRE) != 0; }
    void onChipClearStepSlots() override { if (ocS.hostErr) for (int i = 1; i < 16; ++i) __atomic_store_n(ocS.hostErr + i, 0, __ATOMIC_RELEASE); }
    void onChipRearm(LaunchCtx& ctx) override {
        if (!ocS.bad) return;
        onChipClearStepSlots();
        ocFailed = false; __atomic_store_n(ocS.hostErr, 0, __ATOMIC_RELEASE);
        HIP_CHECK(hipMemsetAsync(ocS.bad, 0, sizeof(int), ctx.stream));
    }
};

template <class T> EnergyOps<T>* makeIW(const unsigned* dims) { return new ImageWarpingOps<T>(dims); }

}  // namespace

EnergyInfo imageWarpingInfo() {
    EnergyInfo e;
    e.name = "image_warping"; e.nDims = 2; e.usePreconditioner = true; e.floatOnly = false; e.residualsPerElement = 10;   // 4 directions x 2 + fit 2 (image_warping.t:13-22)
    e.params = {{ParamDecl::kUnknown, "Offset", "opt_float2", 0}, {ParamDecl::kUnknown, "Angle", "opt_float", 1},
                {ParamDecl::kArray, "UrShape", "opt_float2", 2},  {ParamDecl::kArray, "Constraints", "opt_float2", 3},
                {ParamDecl::kArray, "Mask", "opt_float", 4},      {ParamDecl::kScalar, "w_fitSqrt", "float", 5},
                {ParamDecl::kScalar, "w_regSqrt", "float", 6}};
    e.makeFloat = makeIW<float>; e.makeDouble = makeIW<double>;
    return e;
}

}  // namespace optamd
