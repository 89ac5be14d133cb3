// Type-erased view of the GN/LM solver that the C ABI (opt_api.cpp) drives.
#pragma once
#include "../../include/OptAmd.h"
#include "energy.h"

namespace optamd {

// reference solverGPUGaussNewton.t:148-163 (floats even in double mode), defaults :26-39
struct SolverParameters {
    float min_relative_decrease = 1e-3f;
    float min_trust_region_radius = 1e-32f;
    float max_trust_region_radius = 1e16f;
    float q_tolerance = 0.0001f;
    float function_tolerance = 0.000001f;
    float trust_region_radius = 1e4f;
    float radius_decrease_factor = 2.0f;
    float min_lm_diagonal = 1e-6f;
    float max_lm_diagonal = 1e32f;
    int residual_reset_period = 10;
    int nIter = 0;
    int nIterations = 10;
    int lIterations = 10;
    int patchIterations = 16;     // kind "patchGaussNewtonGPU" only: inner PCG iterations per patch and sweep (reference CUDAPatchSolverWarping.cpp:19)
    int patchSize = 32;           // 16 (the reference's PATCH_SIZE) or 32
    // Path selection (OptAmd.h "Solver parameters of this build"; reference callers never set them -- unknown names only warn there too, solver.t:1205-1221):
    int amd_reference_order = 0;  // 1: the reference's own sequence PCGStep1; PCGStep2; PCGStep3 per PCG iteration (solverGPUGaussNewton.t:1056-1092) on the generic kernels --
                                  // three sums per iteration as the reference forms them (r.z directly, no expansion; r, z, A p in memory).  The loop that meets the 1e-5 contract at
                                  // any horizon; ~3x the bytes of the default single-kernel iteration and never on chip.
    int amd_onchip = 1;           // 0: never take the on-chip (persistent) linear solve; 1: take it where the problem fits (default)
};

struct SolverBase {
    SolverParameters sp;
    KernelTimer timer;
    int verbosity = 0;
    bool traceEnabled = false;
    bool insideSolve = false;    // set by Opt_ProblemSolve around its Init + Step loop: no caller code runs between those steps, so inputs other than the unknowns cannot change
    std::vector<double> trace;   // rows of 6
    virtual ~SolverBase() {}
    virtual void init(void** params) = 0;
    virtual int step(void** params) = 0;
    virtual double cost() const = 0;
    virtual long numUnknownScalars() const = 0;
    virtual void* vector(const std::string& name) = 0;
    virtual void evalJTF(void** params, void* jtf, void* diag) = 0;
    virtual double applyJTJ(void** params, const void* v, void* out) = 0;
    virtual double evalCost(void** params) = 0;
    virtual double trustRegionRadius() const = 0;
    virtual int onChipStatus() const { return 0; }             // OptAmd_PlanOnChipStatus
    virtual std::string describe() { return ""; }              // OptAmd_PlanDescribe
    virtual int setSlab(long row0, long rows, long globalHeight, const OptAmd_SlabComm* comm) = 0;
    virtual int setSlabExt(const OptAmd_SlabCommExt* ext) = 0;
    virtual void setTiming(int mode) = 0;                       // OptAmd_PlanSetTiming: 0 off, 1 an event pair per launch, 2 an event pair per run of launches of one name
    bool setParameter(const char* name, const void* value);   // solver.t:1205-1221
};

SolverBase* makeSolver(const EnergyInfo& info, bool lm, bool patch, bool doublePrecision, const unsigned* dims, bool timing, int verbosity);

}  // namespace optamd
