/* Opt.h -- C ABI of the MI355X-native Gauss-Newton / Levenberg-Marquardt solver backend (libOpt.so).
 *
 * Drop-in boundary: the ten entry points below are exactly the ones niessner/Opt exports from libOpt.a
 * (reference API/release/include/Opt.h:35-71, implemented there by API/src/createwrapper.t:124-220 and
 * API/src/o.t:2521-2558).  Same names, same argument meaning, same ownership rules, same error
 * behaviour (NULL plan on an undefined energy / bad solver kind, abort with a message on a device
 * error).  Behind them there is no Lua VM and no JIT: `filename` selects a hand-written HIP kernel set
 * for gfx950 from a registry, and the solver loop launches those kernels.
 *
 * All image / array / unknown pointers in `problemparams` are DEVICE pointers (hipMalloc'ed or any
 * HIP-visible allocation, e.g. a PyTorch-ROCm tensor's data_ptr); scalar `Param`s and the graph edge
 * count are HOST pointers -- as in the reference (API/src/util.t:664-692).
 */
#pragma once

#ifdef __cplusplus
extern "C" {
#endif

typedef struct Opt_State Opt_State;
typedef struct Opt_Plan Opt_Plan;
typedef struct Opt_Problem Opt_Problem;

/* Set once per Opt_NewState.  A zeroed struct = float, quiet, no timers, 256-thread blocks
 * (reference Opt.h:7-31). */
struct Opt_InitializationParameters {
    /* nonzero: unknowns, opt_float arrays and all solver vectors are double precision.  gfx950 has native
     * f64 atomics, so the graph path does not fall off a cliff in this mode. */
    int doublePrecision;
    /* 0 silent | 1 solver log lines ("cost: a -> b", "final cost=") | 2 + plan/bind diagnostics | 3 same as 2
     * (there is no PTX to dump). */
    int verbosityLevel;
    /* nonzero: bracket every kernel launch with a hipEvent pair and print the per-kernel table at the end
     * of a solve (reference API/src/util.t:451-511). */
    int collectPerKernelTimingInfo;
    /* accepted for compatibility (reference: CUDA block size, multiple of 32, else 256).  The HIP kernels
     * choose wave64-multiple block shapes themselves; the value is recorded and otherwise ignored. */
    int threadsPerBlock;
};
typedef struct Opt_InitializationParameters Opt_InitializationParameters;

/* New independent context (reference Opt.h:35, createwrapper.t:124-211). */
Opt_State* Opt_NewState(Opt_InitializationParameters params);

/* Record the problem specification `filename` (a .t energy file) and the solver kind: "gaussNewtonGPU"
 * or "LMGPU" (reference Opt.h:40, o.t:2521-2525, o.t:122).  Nothing is parsed until Opt_ProblemPlan.
 * Extension of this backend: "patchGaussNewtonGPU" (block-local patch solver, see OptAmd.h). */
Opt_Problem* Opt_ProblemDefine(Opt_State* state, const char* filename, const char* solverkind);
void Opt_ProblemDelete(Opt_State* state, Opt_Problem* problem);

/* Allocate the solver state for `problem` at `dimensions` (indexed by the 2nd argument of each Dim() in
 * the .t).  Returns NULL -- after printing why -- if the energy is not in the kernel registry, its
 * declarations do not match the registered binding layout, or the solver kind is invalid
 * (reference Opt.h:46, o.t:861-882). */
Opt_Plan* Opt_ProblemPlan(Opt_State* state, Opt_Problem* problem, unsigned int* dimensions);
void Opt_PlanFree(Opt_State* state, Opt_Plan* plan);

/* Set a solver parameter by name; `value` points to an int (nIterations, lIterations,
 * residual_reset_period) or a float (the LM knobs) -- also in double mode (reference Opt.h:51,
 * solverGPUGaussNewton.t:148-163, 1205-1221).  Unknown names only warn. */
void Opt_SetSolverParameter(Opt_State* state, Opt_Plan* plan, const char* name, void* value);

/* Init + Step until Step returns 0 (reference Opt.h:56, o.t:2548-2551). */
void Opt_ProblemSolve(Opt_State* state, Opt_Plan* plan, void** problemparams);

/* Bind parameters, reset the iteration counter, evaluate the initial cost (reference Opt.h:62,
 * solverGPUGaussNewton.t:956-1007). */
void Opt_ProblemInit(Opt_State* state, Opt_Plan* plan, void** problemparams);
/* One outer (nonlinear) iteration; 0 = finished (reference Opt.h:65, solverGPUGaussNewton.t:1016-1177).
 * `problemparams` is re-read on every call, so buffers may be swapped between steps. */
int Opt_ProblemStep(Opt_State* state, Opt_Plan* plan, void** problemparams);

/* Cost at the last accepted state (reference Opt.h:70, solverGPUGaussNewton.t:1179-1182). */
double Opt_ProblemCurrentCost(Opt_State* state, Opt_Plan* plan);

#ifdef __cplusplus
}
#endif
