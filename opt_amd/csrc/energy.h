// Interface between the generic GN/LM + PCG solver and one energy's hand-written HIP kernel set.
//
// In the reference this seam is the table of Terra functions Opt's generator emits per energy
// (cost / evalJTF / applyJTJ / computeCtC / modelcost / precompute / exclude -- API/src/o.t:2425-2460)
// which solverGPUGaussNewton.t's kernels call per thread.  Here each energy implements them as whole
// kernels so it can pick its own tiling, LDS staging and reduction shape.
#pragma once
#include "common.h"
#include <string>
#include "../../include/OptAmd.h"

namespace optamd {

// Row-slab description of an image problem tiled over several GPUs (OptAmd_PlanSetSlab).  Local arrays
// hold rows [0,H); rows [yBegin,yEnd) are owned (computed here); local row y is global row gy0+y, which
// exists iff 0 <= gy0+y < Hg.  Single GPU: yBegin=0, yEnd=H, gy0=0, Hg=H.
struct Slab {
    int yBegin = 0, yEnd = 0, gy0 = 0, Hg = 0;
    int ghost = 1;            // ghost rows above and below the owned rows (yBegin == ghost)
    bool active = false;
};

struct UnknownImage {
    int param;        // index in problemparams
    long elems;       // pixels / vertices
    int channels;
    long offset;      // first scalar of this image in the solver's unknown vector
};

// Arguments of the single-kernel PCG iteration (EnergyOps::pcgIteration).  The reference runs three kernels per
// iteration -- PCGStep1 (Ap = A p, alphaDen = p.Ap), PCGStep2 (delta += alpha p, r -= alpha Ap, z = M r,
// betaNum = z.r), PCGStep3 (p = z + beta p) -- separated by two grid-wide sums (solverGPUGaussNewton.t:1056-1091).
// Launch k of the fused kernel does, for every pixel it touches (its rows plus the stencil halo),
//     r_k = r_{k-1} - alpha_{k-1} Ap_{k-1};  z_k = M r_k;  p_k = z_k + beta_{k-1} p_{k-1}          (Step2 + Step3 of iteration k-1)
// then for its own rows   delta += alpha_{k-1} p_{k-1};  Ap_k = A p_k                              (rest of Step2, Step1 of iteration k)
// and the sums  alphaDen_k = p_k.Ap_k,  alphaNum_k = z_k.r_k (direct),  s2 = z_k.Ap_k,  s3 = (M Ap_k).Ap_k.
// beta_{k-1} needs betaNum_{k-1} = sum M r_k^2 before r_k exists; it is obtained from the previous launch's sums by
// expanding the square:  sum M (r - alpha Ap)^2 = alphaNum - 2 alpha s2 + alpha^2 s3  (all sums in double).  Every
// vector update is the reference's; only where that one dot product is evaluated changes.  Old and new r / Ap / p
// buffers must not alias (neighbouring workgroups read the old halo while this one writes).
template <class T>
struct PcgIterArgs {
    const T *rOld, *ApOld, *pOld;
    T *rNew, *ApNew, *pNew;
    T* delta;
    const T* pre;                                        // Jacobi preconditioner (nullptr -> identity)
    int first;                                           // first iteration of a linear solve: alpha_{-1} = beta_{-1} = 0
    Reduction aNumPrev, aDenPrev, s2Prev, s3Prev;        // sums of the previous launch (aNumPrev of launch 0: sum r.p of PCGInit1)
    Reduction *aNum, *aDen, *s2, *s3;                    // sums this launch writes
    // Levenberg-Marquardt (all null / 0 for Gauss-Newton): A = J^T J + diag(CtC); the launch that applies Step2 of iteration k-1 also
    // writes the partial sums of Q_{k-1} = 1/2 sum delta . (r + b) (solver.t:483-485), which the host's q early-out test reads.
    // afterReset: the previous iteration ended with the split residual reset (solver.t:1077-1083), which already updated delta
    // and r; this launch then only forms p = M r + beta p with beta = sum(betaNum) / sum(betaDen) and applies J^T J.
    const T* CtC = nullptr; const T* b = nullptr; Reduction* q = nullptr;
    unsigned qTag = 0;                                   // != 0: q is host-visible and its partials are written as tagged word pairs (common.h storeTaggedPartial)
    int afterReset = 0; Reduction betaNum, betaDen;
    double qInit = 0;                                    // afterReset: Q as the reset's direct sum gave it (a kernel set that carries Q forward by the CG recurrence starts again from it)
    T lmRadius = 0, lmMinDiag = 0, lmMaxDiag = 0;        // the scalars of PCGFinalizeDiagonal (solver.t:631-664): an energy whose diag(J^T J) is a known
                                                         // function of per-pixel flags can rebuild CtC and the LM preconditioner from them instead of reading both
    // Slab mode with a communicator that posts its all-reduces (OptAmd_SlabCommExt.allReducePost): the four sums of the previous launch are not in
    // aNumPrev .. s3Prev but in flight to this rank's mailbox; the kernel's prologue polls them there (mail.words != nullptr; value order aNum, aDen, s2, s3).
    OptAmd_MailRef mail = {nullptr, 0, 0, 0, 0, nullptr};
    // ... and, if the communicator can plan a post (allReducePlan), THIS launch posts its own four sums from its last workgroup (post.world != 0): no kernel
    // of the communicator's between two iterations.
    OptAmd_MailPost post = {};
    T* deltaOut = nullptr;                               // if set, the updated delta goes here instead of in place (lets the solver enqueue
                                                         // the next launch before it has read Q: an early-out then still finds the old delta)
};

// Arguments of EnergyOps::evalJTFInitLM: what PCGInit1 + PCGSaveSSq + PCGFinalizeDiagonal (solver.t:361-419, 624-664; solver.hip k_finalizeDiagonal<T, true>) read and write.
template <class T>
struct LmInitArgs {
    T *CtC, *SSq, *r, *delta, *pre, *b, *p;
    T radius, minLm, maxLm;
    int saveSSq;                    // first outer iteration: SSq <- guardedInvert(diag) (or of 1 without preconditioner)
    Reduction *rDotP, *q;           // partial sums of r . p (the first alphaNumerator) and of Q_0 (exactly 0)
};

// Levenberg-Marquardt controls of EnergyOps::pcgSolveOnChip: the scalars of PCGFinalizeDiagonal (solver.t:631-664), q_tolerance and residual_reset_period (:1077-1102).
template <class T>
struct OnChipLm { T radius, minLm, maxLm, qTolerance; int resetPeriod; const T* CtC = nullptr;
                  double* breakInfo = nullptr; };      // pinned, 2 doubles: workgroup 0 leaves {iteration of the q early-out + 1, zeta} there (0: the loop ran to its end) -- the solver prints the
                                                       // reference's "breaking at iteration" message from it when someone listens (verbosity > 0)      // CtC: the clamped diagonal PCGFinalizeDiagonal has just written (energies whose kernel does not rebuild it from a table)

// Everything the solver needs from an energy.  T = opt_float (float or double).
// Contract shared by all implementations:
//  * solver vectors are laid out like the unknown vector: unknown images in declaration order, AoS,
//    concatenated (reference o.t:675-687, 745-775);
//  * rows of excluded unknowns (reference Exclude(), o.t:2452-2455) and of ghost rows are written as 0 by
//    evalJTF (r and diag) and applyJTJ (out), so the generic streaming kernels never need the mask:
//    with r = 0 there, p, z, delta stay 0 and X is unchanged, which is what the reference's
//    "if not exclude" guards achieve (solverGPUGaussNewton.t:371, 424, 450, 539, 554).
template <class T>
struct EnergyOps {
    std::vector<UnknownImage> unknowns;
    long nScalars = 0;
    bool usePreconditioner = false;   // reference o.t:214 default
    bool usesGraph = false;
    Slab slab;
    bool iterStateExchange = false;   // set by pcgIteration: in slab mode the solver exchanges the ghost rows of r and p after a launch (else of Ap before it)
    bool iterExchangeDue = true;      // ... and whether that exchange is needed after THIS launch (deep ghost zones let a kernel skip some)
    bool iterTakesMail = false;       // set by pcgIteration: its kernels can read the previous launch's sums from a posted all-reduce (PcgIterArgs::mail)
    // Before the first launch of a loop: would pcgIteration's kernels carry out a planned post themselves and poll the mailbox (PcgIterArgs::post / mail)?
    virtual bool iterPostsItself(bool /*lm*/) const { return false; }
    virtual ~EnergyOps() {}
    void addUnknown(int param, long elems, int channels) {
        unknowns.push_back({param, elems, channels, nScalars});
        nScalars += elems * channels;
    }
    // util.initParameters (reference util.t:664-692): re-read every pointer and host scalar; also refresh
    // any per-solve auxiliary arrays derived from the inputs (validity flags ...).
    virtual void bind(void** params, LaunchCtx& ctx) = 0;
    // Inside Opt_ProblemSolve (Init + Steps with no caller code in between) the reference re-reads the same pointers and host scalars before every step (util.t:664-692): a
    // kernel set whose bind() derives device-side auxiliaries ONLY from inputs that are not unknowns (flag images, edge lists ...) may say so here, and the solver binds once
    // per solve instead of once per step.  Opt_ProblemStep called by itself always binds.
    virtual bool bindInvariantDuringSolve() const { return false; }
    // pcgIteration kept the search directions in buffers of its own: where p of the launch issued last lives (nullptr: in the caller's pNew)
    virtual const T* iterCurrentP() const { return nullptr; }
    // LM loop: did the launch issued last write deltaOut (a kernel set that pairs its delta updates writes it every second launch) ...
    virtual bool iterWroteDelta() const { return true; }
    // ... and the term such a launch left owed, added to `delta` in place (issuedBeyond: launches issued after the one meant -- the solver's speculative next launch)
    virtual void iterFlushDelta(T* /*delta*/, int /*issuedBeyond*/, LaunchCtx&) {}
    // ... or handed to a solver kernel that adds it in its own pass: *p, *alpha[0] (false: nothing owed)
    virtual bool iterOwedTerm(int /*issuedBeyond*/, const T** /*p*/, const T** /*alpha*/) const { return false; }
    virtual T* unknownPtr(int img) const = 0;
    virtual void precompute(LaunchCtx&) {}                                   // ComputedArrays (solver.t:607-614)
    // partial sums of 1/2 sum r^2 over non-excluded, owned elements (solver.t:580-592, 715-725)
    virtual void evalCost(Reduction& out, LaunchCtx& ctx) = 0;
    // r = -J^T F, diag = diag(J^T J) (raw, also when the energy does not precondition) (o.t:2129-2172, 2228-2253)
    virtual void evalJTF(T* r, T* diag, LaunchCtx& ctx) = 0;
    // out = J^T J v (+ CtC .* v if CtC != nullptr, o.t:2076-2082); dot (optional) = partial sums of v . out
    virtual void applyJTJ(const T* v, T* out, const T* CtC, Reduction* dot, LaunchCtx& ctx) = 0;
    // Optional fusion of the previous iteration's PCGStep3 into PCGStep1: pNew = z + beta pOld with
    // beta = sum(bNum) / aNumOld (guarded, solver.t:544-547), aNumNext[0] = sum(bNum), then out = J^T J pNew.
    // pNew must not alias pOld.  Return false if the energy has no fused kernel (the solver then runs
    // the generic PCGStep3 followed by applyJTJ).
    virtual bool applyJTJFused(const T* /*pOld*/, const T* /*z*/, T* /*pNew*/, T* /*out*/, const T* /*CtC*/, Reduction* /*dot*/,
                               const Reduction& /*bNum*/, const double* /*aNumOld*/, double* /*aNumNext*/, LaunchCtx&) { return false; }
    // Optional (Gauss-Newton, single GPU): PCGInit1 and PCGInit1_Finish (solver.t:361-419) in one go -- r = -J^T F, p = M r with the guarded-inverse Jacobi
    // preconditioner, delta = 0, aNum0 = partial sums of r.p -- for a kernel set whose pcgIteration needs neither the diag nor the preconditioner vector.
    // Returning true promises that pcgIteration will accept the loop that follows.  false: the solver runs evalJTF + its own flat pass.
    virtual bool evalJTFInit(T* /*r*/, T* /*p*/, T* /*delta*/, long /*nPad*/, Reduction& /*aNum0*/, LaunchCtx&) { return false; }
    // computeCost of the step that has just updated the unknowns and evalJTFInit of the next step in one pass over them (both read the same X; the cost's partial sums
    // are those of evalCost: same grid, same expressions).  Only asked for inside Opt_ProblemSolve, where no caller code runs between the two steps.
    virtual bool evalCostAndJTFInit(Reduction& /*cost*/, T* /*r*/, T* /*p*/, T* /*delta*/, long /*nPad*/, Reduction& /*aNum0*/, LaunchCtx&) { return false; }
    // Optional (Levenberg-Marquardt, single GPU): PCGInit1 and everything k_finalizeDiagonal<T, true> does after it, in the energy's own J^T F kernel -- r = -J^T F,
    // CtC = clamped diag(J^T J) / radius, the LM preconditioner, b = r, p = M r, delta = 0, SSq (first outer iteration), partial sums of r . p.  false: the solver
    // runs evalJTF and its flat pass.
    virtual bool evalJTFInitLM(const LmInitArgs<T>& /*args*/, LaunchCtx&) { return false; }
    // LM's split residual reset behind delta += alpha p (solver.t:1079-1083): Adelta = (J^T J + CtC) delta and r = b - Adelta, z = M r, the partial sums of r . z and of
    // Q = 1/2 delta . (r + b) -- in one pass where the kernel set can (false: applyJTJ + the solver's k_step2SecondHalf)
    virtual bool applyJTJResetLM(const T* /*delta*/, T* /*r*/, const T* /*b*/, const T* /*pre*/, T* /*z*/, const T* /*CtC*/, Reduction& /*bNum*/, Reduction& /*q*/, LaunchCtx&) { return false; }
    // Optional: the end of a single-kernel Gauss-Newton loop in one pass over the unknowns -- whatever pcgFinish would still add to delta, the last iteration's
    // delta += alpha p (alpha = sum aNum / sum aDen, guarded) and PCGLinearUpdate X += delta.  delta itself is dead afterwards and need not be written.
    // true: the unknowns are updated (the solver skips pcgFinish, the last PCGStep2 and PCGLinearUpdate).
    virtual bool finishUpdate(const T* /*pPrev*/, const T* /*pLast*/, const T* /*delta*/, const Reduction& /*aNum*/, const Reduction& /*aDen*/, LaunchCtx&) { return false; }
    // Optional: one WHOLE Gauss-Newton PCG iteration as a single kernel (see PcgIterArgs and solver.hip).
    virtual bool pcgIteration(const PcgIterArgs<T>& /*args*/, LaunchCtx&) { return false; }
    // Optional (single GPU; Gauss-Newton also on row slabs): the WHOLE linear solve -- lIterations PCG iterations (solverGPUGaussNewton.t:1056-1092) from r = r_0, p = M r_0 as PCGInit1
    // left them, then PCGLinearUpdate X += delta -- as one persistent launch that keeps the loop state on chip (iw_onchip.h).  r0 and p0 are only read; delta
    // receives sum alpha_k p_k; traceDev (or nullptr) receives alphaNum, alphaDen, s2, s3 of every iteration (4 doubles each; beta numerator by expansion as
    // in PcgIterArgs).  false: the problem does not fit the chip or the kernel set has no such kernel -- nothing was touched.
    // lm != nullptr: the Levenberg-Marquardt loop instead (A = J^T J + diag(CtC), the q early-out decided on chip, the split residual reset as a second stencil pass;
    // r0 = b and p0 as PCGFinalizeDiagonal left them).  Then the kernel only produces delta: the solver applies savePreviousUnknowns + PCGLinearUpdate itself.
    virtual bool pcgSolveOnChip(const T* /*r0*/, const T* /*p0*/, T* /*delta*/, int /*lIterations*/, double* /*traceDev*/, const OnChipLm<T>* /*lm*/, LaunchCtx&) { return false; }
    // an energy that does not precondition (UsePreconditioner(false): no preconditioner vector exists) but offers pcgSolveOnChip all the same
    virtual bool onChipWithoutPreconditioner() const { return false; }
    // Gauss-Newton: did pcgSolveOnChip end with PCGLinearUpdate (X += delta) itself?  false: the solver applies delta as after any other linear solve
    virtual bool onChipAppliedUpdate() const { return true; }
    // ... and if it did not: PCGLinearUpdate X += delta applied by the kernel set itself, GUARDED by the launch's failure word (a workgroup that gave up in the last wait only
    // raises the flag while the others have already written their delta: an unguarded update would add a partial delta that Gauss-Newton -- no saved unknowns -- cannot take
    // back).  false: no such kernel, the solver applies delta itself (Levenberg-Marquardt restores the saved unknowns on a failure).
    virtual bool onChipGuardedUpdate(const T* /*delta*/, LaunchCtx&) { return false; }
    // Row slabs: would pcgSolveOnChip run for this rank's slab right now (kernel variant fits, unit lattice, the communicator offers onChipPlan ...)?  The solver
    // makes the decision collective (all ranks or none) before anyone launches.  onChipPlan / onChipCtx: the communicator's entry (OptAmd_SlabCommExt), set by the solver.
    virtual bool slabOnChipAvailable(int /*lIterations*/) { return false; }
    int (*onChipPlan)(void*, int, int, int, long, OptAmd_OnChipLinks*) = nullptr;
    void* onChipCtx = nullptr;
    // Row slabs, behind a pcgSolveOnChip launch (which then applies nothing itself): onChipVerdict leaves this rank's verdict (0 fine / 1 failed) in a device scalar, the
    // solver all-reduces it, onChipApply applies X += delta iff the sum is 0 -- every rank keeps its update or none does -- and tells the host (onChipFailed).
    virtual void onChipVerdict(double* /*out*/, bool /*refused*/, LaunchCtx&) {}
    virtual void onChipApply(const T* /*delta*/, const double* /*verdict*/, bool /*refused*/, LaunchCtx&) {}
    // OptAmd_PlanDescribe: which linear-solve path the kernel set would take for the plan as it stands (dims, slab), as "key=value; ..." -- bench.py --dry prints it per
    // rank so that a multi-GPU run can be read before it is started.
    virtual std::string describe(int /*lIterations*/, bool /*lm*/) { return "path=launch-per-iteration"; }
    // After the stream has drained: did a wait inside the last on-chip solve time out (another tenant on the GPU kept its workgroups from being co-resident)?
    // Then the unknowns were left untouched, the kernel set has switched the path off for this plan, and the caller redoes the linear solve.
    virtual bool onChipFailed() { return false; }
    // the same question without consuming the answer (the solver prints an on-chip LM solve's "breaking at iteration" message only for a launch that did not fail)
    virtual bool onChipFailedPeek() { return false; }
    // The solver's back-off after such a failure is over: clear the failure state (device word, pinned word, the path's own off switch) so that the next
    // pcgSolveOnChip launches again.  Stream-ordered.
    virtual void onChipRearm(LaunchCtx&) {}
    // Opt_ProblemSolve may enqueue several Gauss-Newton steps before it reads anything back (PcgSolver: deferred steps).  A kernel set that supports it gives every such
    // step's guarded update a word of its own (slot >= 0; -1: none), so that the host can tell afterwards WHICH step's on-chip solve gave up -- from that step on nothing was
    // applied (the failure flag is sticky until onChipRearm) and the solver goes back to it.
    // pcgIteration takes delta from its arguments at every launch and keeps no pointer to it: the solver may move the vector between two launches (PcgSolver::deltaTrial)
    virtual bool deltaMovable() const { return false; }
    virtual bool supportsDeferredSteps() const { return false; }
    virtual void onChipStepSlot(int /*slot*/) {}
    virtual bool onChipStepFailed(int /*slot*/) { return false; }
    virtual void onChipClearStepSlots() {}
    // Slab mode, before the loop: will pcgIteration accept the launches?  (The solver refreshes the ghost rows of r_0, p_0 and M for that loop only: the
    // three-kernel loop relies on r being 0 on ghost rows -- its flat sums run over them.)
    virtual bool slabIterationAvailable() const { return true; }
    // Slab mode, after a pcgIteration launch with iterStateExchange: which vectors (solver layout) carry the state whose ghost rows the neighbours
    // must refresh.  0 = the rNew / pNew the launch was given; a kernel set that keeps its loop state in buffers of its own lists them here.
    virtual int iterExchangeVectors(T** /*out4*/) { return 0; }
    // Called once after the last pcgIteration of a linear solve, before the solver adds the last term alpha p to delta:
    // lets a kernel set that defers part of its delta update apply what is left.  pPrev = the p buffer the last launch read.
    // Returns where the search direction of the last launch lives if the kernel set kept it in a buffer of its own (nullptr: in the pNew it was given).
    virtual const T* pcgFinish(const T* /*pPrev*/, T* /*delta*/, LaunchCtx&) { return nullptr; }
    // Optional block-local solver (kind "patchGaussNewtonGPU", OptAmd.h): one additive-Schwarz sweep of LDS-resident patch PCG solves over
    // a tiling shifted by (fx, fy) patch widths, nPatchIters inner iterations each, applied to the unknowns directly; patchFinish is called
    // after the last sweep of a step and must leave the result in the caller's unknown buffers.  false = the energy has no such kernel.
    virtual bool patchIteration(float /*fx*/, float /*fy*/, int /*nPatchIters*/, int /*patchSize*/, LaunchCtx&) { return false; }
    virtual void patchFinish(LaunchCtx&) {}
    virtual bool supportsPatch() const { return false; }
    // partial sums of 1/2 sum (F + J delta)^2 (o.t:2174-2225); LM only
    virtual void evalModelCost(const T* delta, Reduction& out, LaunchCtx& ctx) = 0;
    // slab tiling (image energies): number of scalars in one image row of unknown image `img`
    virtual bool supportsSlab() const { return false; }
    virtual long rowScalars(int /*img*/) const { return 0; }
};

// Registry entry: what a .t file must declare for this kernel set, and how to instantiate it.
struct ParamDecl {
    enum Kind { kUnknown, kArray, kScalar, kGraphCount, kGraphIndex } kind;
    const char* name;
    const char* type;   // "opt_float", "opt_float2", ..., "float", "uint8", "int"
    int index;          // binding index in problemparams
};
struct EnergyInfo {
    const char* name;                 // .t file stem
    int nDims;                        // entries of `dimensions` consumed
    std::vector<ParamDecl> params;
    bool usePreconditioner;
    bool floatOnly;                   // energy declares fixed `float` unknowns (tests/minimal/laplacian.t)
    int residualsPerElement = 0;      // scalar residuals per element of the index space / per graph edge (plan-time report, solverGPUGaussNewton.t:128-142); 0 = not stated
    int residualsPerEdge = 0;
    EnergyOps<float>* (*makeFloat)(const unsigned* dims);
    EnergyOps<double>* (*makeDouble)(const unsigned* dims);
};
const std::vector<EnergyInfo>& energyRegistry();

}  // namespace optamd
