/* OptAmd.h -- extensions of libOpt.so beyond the reference's Opt.h.
 *
 * The reference has no equivalent of these: its only per-vector probe is the hand-written comparator's
 * dump kernels (examples/shape_from_shading/src/SFSSolver.cu:409-463) and its only timing output is the
 * printed table (API/src/util.t:451-511).  They exist so that (a) parity tests can compare every
 * intermediate vector of the HIP solver with the CPU oracle, (b) bench.py can read per-kernel hipEvent
 * timings for the roofline figure, (c) a multi-process launcher can tile an image problem over the GPUs
 * of a node.  A caller that only needs the reference behaviour never touches this header.
 *
 * One extension has no entry point of its own: Opt_ProblemDefine accepts a third solver kind,
 * "patchGaussNewtonGPU" -- the block-local patch solver that the reference ships as a separate CUDA solver
 * of its poisson example (examples/poisson_image_editing/src/PatchSolverWarping.cu:67-241, driven by
 * CUDAPatchSolverWarping.cpp:14-33).  Opt_ProblemPlan returns NULL for energies without a patch kernel.
 * Opt_SetSolverParameter names: "nIterations" (outer steps), "lIterations" (sweeps per step; the tiling
 * is shifted by the next Halton point before each), "patchIterations" (int, PCG iterations inside a patch,
 * default 16), "patchSize" (int, 16 or 32, default 32).  Init / Step / CurrentCost behave as for the other kinds.
 */
#pragma once
#include "Opt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Library identification; the string names the compile target ("gfx950"). */
const char* OptAmd_Version(void);
/* Kernel registry: energies (".t" file stems) this build has hand-written kernel sets for. */
int OptAmd_EnergyCount(void);
const char* OptAmd_EnergyName(int i);

/* Host-only check of a .t file against the registry (what Opt_ProblemPlan does before touching the GPU):
 * returns 1 and "ok: <energy>" in `message`, or 0 and the reason Opt_ProblemPlan would print. */
int OptAmd_CheckProblemFile(const char* filename, char* message, int messageLen);

/* FNV-1a hash of the comment- and whitespace-stripped text of a .t file: Opt_ProblemPlan accepts an energy only if this hash is one of
 * the body versions its hand-written kernel set implements (the reference compiles whatever the file says; this backend cannot, so an
 * edited Energy / Exclude body is refused instead of silently solved as the unedited energy).  0 if the file cannot be read. */
unsigned long OptAmd_ProblemFileHash(const char* filename);

/* Length of the solver's unknown vector: unknown images in declaration order, each AoS, concatenated
 * (the reference's UnknownType iteration order, API/src/o.t:675-687). */
long OptAmd_PlanNumUnknownScalars(Opt_Plan* plan);
/* Device pointer of a solver vector owned by the plan: "delta","r","b","Adelta","z","p","Ap_X","CtC",
 * "preconditioner","SSq","prevX" (reference PlanData fields, solverGPUGaussNewton.t:173-185).
 * NULL for an unknown name.
 * What they hold after an Opt_ProblemStep is the reference's content only where the path that ran materialises the vector.  Gauss-Newton image_warping on
 * one GPU keeps its loop state elsewhere: "delta" holds the step's update only when the linear solve ran on chip (iw_onchipPcg); with the streaming loop the
 * last one or two alpha * p terms go straight into the unknowns and "delta" lacks them (it is never written when lIterations <= 2); "p" / "r" hold the
 * PCGInit1 values, "preconditioner" / "CtC" / "Ap_X" are not written at all.  Every other energy, LM, row slabs and OPT_AMD_ONEKERNEL=0 fill them as
 * the reference does. */
void* OptAmd_PlanVector(Opt_Plan* plan, const char* name);

/* Kernel-level entry points.  Each binds `problemparams` exactly like Opt_ProblemInit, launches the
 * energy's kernel(s) and synchronises.  Outputs are DEVICE buffers of OptAmd_PlanNumUnknownScalars
 * opt_float elements.
 *   EvalJTF : jtf = J^T F (gradient without the factor 2), diag = diag(J^T J); rows of excluded
 *             unknowns are 0 (reference evalJTF, o.t:2129-2172, 2228-2253).
 *   ApplyJTJ: out = J^T J v on non-excluded rows, 0 elsewhere; returns v^T J^T J v (the PCG alpha
 *             denominator) (reference applyJTJ, o.t:2029-2126, called by PCGStep1 solver.t:421-434).
 *   EvalCost: 1/2 sum r^2 over non-excluded elements (reference cost, o.t:2375-2385). */
void OptAmd_EvalJTF(Opt_State* state, Opt_Plan* plan, void** problemparams, void* jtf, void* diag);
double OptAmd_ApplyJTJ(Opt_State* state, Opt_Plan* plan, void** problemparams, const void* v, void* out);
double OptAmd_EvalCost(Opt_State* state, Opt_Plan* plan, void** problemparams);

/* Per-PCG-iteration scalars of the solve since the last Opt_ProblemInit.  Recording costs one device->host
 * read per PCG iteration, so it is off unless enabled.  Row = {nIter, lIter, alphaNumerator,
 * alphaDenominator, betaNumerator, q}. */
void OptAmd_PlanEnableTrace(Opt_Plan* plan, int enable);
long OptAmd_PlanTraceRows(Opt_Plan* plan);
void OptAmd_PlanGetTrace(Opt_Plan* plan, double* rows6);

/* Current LM trust-region radius (reference pd.parameters.trust_region_radius). */
double OptAmd_PlanTrustRegionRadius(Opt_Plan* plan);

/* Which linear-solve path the plan's last step took:
 *   0  launch-per-iteration kernels (the problem does not fit the chip, the kernel set has no on-chip solve, or it is switched off: "amd_onchip" / "amd_reference_order");
 *   1  the whole linear solve of the last step ran as one persistent on-chip launch;
 *   2  the plan is in its back-off after a failed on-chip launch: the waits of a launch's first phase are bounded by 10 ms (passing them proves the whole grid resident;
 *      a foreign tenant holding CUs makes the launch give up there, before anything has been written), the step was redone by the streaming kernels (reported on stderr the
 *      first three times) and the plan stays on them for 8 (then 16, 32 ... 1024) steps before it tries the chip again.  OptAmd_PlanDescribe carries `onchip_fallbacks` and
 *      `onchip_backoff_steps_left`.  Plans of one process stepped from several host threads take turns on the chip (a per-device lease around each on-chip launch).
 *      The two paths round differently: a caller that compares runs should check this. */
int OptAmd_PlanOnChipStatus(Opt_Plan* plan);
/* What the plan WOULD do at its next step, as "key=value; ..." text (truncated to outLen - 1 characters; returns the full length): the linear-solve path of its kernel
 * set for the plan's dimensions and slab (on chip or one launch per iteration, and why), tiles, LDS, ghost depth, bytes that cross ranks per PCG iteration, the
 * communicator's fast paths.  Uses the solver parameters set so far (lIterations).  `bench.py --gpus N --dry` prints it per rank without running a step. */
int OptAmd_PlanDescribe(Opt_Plan* plan, char* out, int outLen);

/* Solver parameters of this build, set through the reference's own Opt_SetSolverParameter (a reference build only warns about names it does not know,
 * solverGPUGaussNewton.t:1205-1221, so a caller that sets them stays portable); both are `int*`:
 *   "amd_reference_order"  1: every PCG iteration runs the reference's own sequence PCGStep1; PCGStep2; PCGStep3 (solverGPUGaussNewton.t:1056-1092) on kernels that keep
 *                          r, z and A p in memory and form the three sums as the reference does (beta numerator = r.z directly).  This is the loop that meets the 1e-5
 *                          (float) contract against the oracle at every horizon tested (400 PCG iterations included); it moves the reference formulation's 180 B/pixel
 *                          per iteration where the default single-kernel iteration moves 53 and it never runs on chip.  0 (default): the fused loops.
 *   "amd_onchip"           0: never take the on-chip (persistent) linear solve; 1 (default): take it where the problem fits the chip.
 * OptAmd_PlanDescribe reports the choice.  (The environment switches OPT_AMD_ONEKERNEL / OPT_AMD_ONCHIP remain as process-wide development overrides.) */

/* The float4 copy rate of this box in GB/s: `bytes` moved in total per repetition (half read, half written; device memory allocated and freed inside the call),
 * default (0) or nontemporal (1) accesses, `reps` timed repetitions after three warm-up launches.  bench.py reports it next to its roofline fraction: boxes of the
 * pool differ by ~10 % and MI355X_MICROARCH.md's copy ceiling (6.29 TB/s) is a measurement of this kind.  0.0 if the buffers cannot be allocated. */
double OptAmd_MeasureCopyBandwidth(long bytes, int nontemporal, int reps);
/* Test hook: `workgroups` one-wave workgroups, each holding 150 KB of LDS (at most one per CU), that spin for `milliseconds` (<= 5000) on `stream` (a hipStream_t) -- a
 * stand-in for a foreign tenant holding that many CUs while a
 * plan's persistent kernel is launched (tests/test_coresidency_gpu.py).  Returns 1 if launched. */
int OptAmd_DebugOccupy(int workgroups, double milliseconds, void* stream);

/* hipEvent timing of one kernel name since the last Opt_ProblemInit (requires
 * collectPerKernelTimingInfo).  Returns 0 if the name was never launched. */
int OptAmd_PlanKernelTiming(Opt_Plan* plan, const char* kernel, long* count, double* total_ms);
/* Switch the per-kernel hipEvent timing of a plan on or off between steps (what
 * Opt_InitializationParameters.collectPerKernelTimingInfo fixes at plan time; the reference has no such call).
 * The totals restart from zero.  bench.py times its steps without the events and then turns them on for the
 * roofline leg of the same job.  enable = 2: one event pair per RUN of consecutive launches under the same name (the start of the first, the
 * end of the last) instead of one per launch: a loop of 400 PCGIteration launches costs two event records and its time is the loop's own. */
void OptAmd_PlanSetTiming(Opt_Plan* plan, int enable);
/* Number of distinct kernel names timed, and the i-th name. */
int OptAmd_PlanKernelCount(Opt_Plan* plan);
const char* OptAmd_PlanKernelName(Opt_Plan* plan, int i);

/* ---- multi-GPU tiling of image problems (one process per GPU) -------------------------------------
 * The image is split into contiguous row slabs; this process owns rows [row0, row0+rows) of an image of
 * global height `globalHeight` and passes arrays that hold its slab PLUS one ghost row above and below
 * (rows row0-1 .. row0+rows; ghost rows outside the global image are ignored).  Opt_ProblemPlan is then
 * called with dims = {W, rows+2}.  The three callbacks carry the only inter-GPU traffic of the path:
 * a 1-row halo exchange of a solver vector / unknown image and a sum all-reduce of a few doubles.  They
 * are invoked on the solver's stream (passed as `stream`, a hipStream_t) and must be stream-ordered
 * (RCCL calls enqueued on that stream, or host code that synchronises it). */
typedef struct OptAmd_SlabComm {
    void* ctx;
    int rank, world;
    /* One halo exchange of `nBuffers` row buffers (one per unknown image): for each k send bytes[k] bytes from
     * sendUp[k] to rank-1 and from sendDown[k] to rank+1, receive into recvUp[k] (from rank-1) and recvDown[k]
     * (from rank+1).  Sides without a neighbour (rank 0 / rank world-1) must be skipped by the callee. */
    void (*haloExchange)(void* ctx, int nBuffers, const void* const* sendUp, const void* const* sendDown, void* const* recvUp, void* const* recvDown,
                         const long* bytes, void* stream);
    /* in-place sum all-reduce of n doubles in device memory */
    void (*allReduceSum)(void* ctx, double* deviceBuf, int n, void* stream);
} OptAmd_SlabComm;

/* Where a device-side consumer finds the result of a POSTED all-reduce (OptAmd_SlabCommExt.allReducePost): a mailbox in this rank's own device memory
 * into which every rank (this one included) stores its contribution as self-validating 8-byte words -- word [source rank * stride + 2 * value + half]
 * = (tag << 32) | 32 bits of the double's payload, each stored whole.  The consumer polls until all `world` x 2n words carry `tag` and adds the
 * contributions in rank order (the same bits on every rank).  Polls are bounded by `timeoutTicks` of the device's wall clock (100 MHz); on expiry the
 * consumer stores 1 to *errFlag (pinned host memory, checked by the communicator at its next call) and continues with NaN. */
typedef struct OptAmd_MailRef {
    const unsigned long long* words;
    int world, stride;
    unsigned tag;
    long long timeoutTicks;
    int* errFlag;
} OptAmd_MailRef;

/* The producer side of a posted all-reduce when the PRODUCER KERNEL performs the post itself (OptAmd_SlabCommExt.allReducePlan): the kernel's workgroups
 * write their partial sums as usual, take a ticket from *ticket (device-scope atomic add, zero when the kernel starts); the workgroup that draws the last
 * one adds up all partials (256 threads, the order of the communicator's own post kernel) and stores value i of its n sums as the two tagged words
 * dst[t][2 i], dst[t][2 i + 1] = (tag << 32) | payload half into the mailbox of every rank t < world, then resets *ticket to 0. */
typedef struct OptAmd_MailPost {
    unsigned long long* dst[16];
    int world;
    unsigned tag;
    unsigned* ticket;
} OptAmd_MailPost;

/* Links of the on-chip linear solve across ranks (OptAmd_SlabCommExt.onChipPlan): a persistent kernel on every rank runs `count` PCG iterations; in iteration k
 * (0-based) its workgroup 0 stores the rank's n sums as tagged words -- value i as mailDst[t][slot * slotStride + 2 i], [.. + 2 i + 1] = (tag << 32) | payload half,
 * slot = (seq0 + k) % slots, tag = seq0 + k -- into the mailbox of every rank t < world, and every workgroup polls mailMine[slot * slotStride + r * rankStride + w] for
 * all r < world.  The edge tiles of neighbouring slabs hand each other rows of tagged words: this rank's top tiles store into edgeSendUp (the lower edge box of the
 * rank above) and poll edgeRecvUp; likewise Down.  An edge box holds [parity 2][tile][wordsPerTile] words, tagged like the mailbox words with tag = seq0 + k
 * (parity = tag & 1: a communicator-wide sequence number, so no word left by an earlier plan can match); NULL where there is no neighbour.
 * All stores / loads are relaxed, system-scope, 8 bytes.  Polls are bounded by timeoutTicks (100 MHz); on expiry the kernel stores a non-zero code to *errFlag. */
typedef struct OptAmd_OnChipLinks {
    unsigned long long* mailDst[16];
    const unsigned long long* mailMine;
    int world, rank, slots, slotStride, rankStride;
    unsigned seq0;
    unsigned long long* edgeSendUp; unsigned long long* edgeSendDown;
    const unsigned long long* edgeRecvUp; const unsigned long long* edgeRecvDown;
    long edgeParityStride;
    long long timeoutTicks;
    int* errFlag;
} OptAmd_OnChipLinks;

/* Optional accelerations of a communicator; any member may be NULL.  `size` must be sizeof(OptAmd_SlabCommExt) as the caller compiled it (a library
 * built against a longer struct reads no member beyond it), and the struct must be zero-initialised before the members are set. */
typedef struct OptAmd_SlabCommExt {
    unsigned long size;
    /* The all-reduce of allReduceSum fed with per-workgroup partial sums -- value i = sum over ranks of sum(partials[i][0 .. counts[i])) -- written to
     * out[0 .. n) on every rank; lets an implementation fold the local reduction into its own kernel (one launch between two PCG iterations instead of
     * two).  n <= 8. */
    void (*allReducePartials)(void* ctx, const double* const* partials, const int* counts, int n, double* out, void* stream);
    /* The same sums, POSTED only: the local reduction and the stores into every rank's mailbox are enqueued on `stream`, nobody waits; *ref tells a
     * kernel enqueued behind it where to poll (OptAmd_MailRef).  The PCG iteration kernel then starts -- launch latency, first row loads -- while the
     * contributions are still crossing the links, instead of behind a kernel that waited for them.  Returns 0 if unavailable.  n <= 8.  At most two
     * posted all-reduces may be outstanding (the mailbox has four slots). */
    int (*allReducePost)(void* ctx, const double* const* partials, const int* counts, int n, OptAmd_MailRef* ref, void* stream);
    /* CO-RESIDENCY: every workgroup of the consuming kernel polls the mailbox, so the kernels of ALL ranks must be resident at the same time -- true with one GPU
     * per rank, false for ranks that share a GPU with full-chip grids (a post kernel then queues behind a peer's launch that is waiting for it).  A communicator
     * must leave this member (and allReducePlan) NULL when ranks share a device; libOptComm's peer communicator does (it compares PCI bus ids at connect time). */
    /* The same without any kernel of the communicator's: reserves the next all-reduce and describes it -- *post for the kernel that produces the n sums
     * (it posts them itself from its last workgroup, OptAmd_MailPost), *ref for the kernel that consumes them.  Between two PCG iteration kernels there is
     * then nothing at all.  Returns 0 if unavailable.  The reserved all-reduce MUST be carried out by a kernel enqueued before the next call of any entry point
     * of this communicator. */
    int (*allReducePlan)(void* ctx, int n, OptAmd_MailPost* post, OptAmd_MailRef* ref);
    /* Reserve `count` consecutive all-reduces of n doubles and the edge boxes for tilesX tiles of wordsPerTile words per slab edge, for ONE persistent kernel per rank
     * that carries all of them out itself (OptAmd_OnChipLinks).  Same co-residency requirement as allReducePost, over the whole launch.  Returns 0 if unavailable
     * (too many tiles for the boxes, ranks sharing a device, ...): the caller then runs its streaming loop.  Every rank must make the same call.
     * count == 0 (links may be NULL) is a dry query: 1 if such a plan could be made right now, nothing reserved -- a rank asks it before it votes for running on chip. */
    int (*onChipPlan)(void* ctx, int n, int count, int tilesX, long wordsPerTile, OptAmd_OnChipLinks* links);
} OptAmd_SlabCommExt;
/* Attach a slab description to a plan created with dims {W, rows + 2*g}: g >= 1 ghost rows above and below the `rows` owned
 * rows (g is inferred from the plan's height).  g = 1 is enough for every kernel set; with g >= 2 image_warping runs its
 * PCG iteration without the A*p vector in memory (the neighbours' g edge rows of r and p are exchanged instead of one
 * row of A*p), and that exchange is needed only every g - 1 iterations because the kernel keeps the ghost rows it still
 * needs current by itself.  Must precede Opt_ProblemInit. */
int OptAmd_PlanSetSlab(Opt_Plan* plan, long row0, long rows, long globalHeight, const OptAmd_SlabComm* comm);
/* Optional, after OptAmd_PlanSetSlab: hand the plan a communicator's accelerated entry points (same ctx as the OptAmd_SlabComm).  Returns 1 if accepted. */
int OptAmd_PlanSetSlabExt(Opt_Plan* plan, const OptAmd_SlabCommExt* ext);

#ifdef __cplusplus
}
#endif
