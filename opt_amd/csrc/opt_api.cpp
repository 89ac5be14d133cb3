// The C ABI of libOpt.so: Opt.h's ten entry points + the OptAmd.h extensions.
//
// Replaces reference API/src/createwrapper.t (per-state Lua VM forwarding to Terra function pointers,
// :85-220) and the API functions of API/src/o.t:2521-2558.  No VM here: a state is a small struct, a
// problem is {filename, kind}, a plan owns a solver object bound to one energy's HIP kernel set.
#include "../../include/Opt.h"
#include "../../include/OptAmd.h"
#include "solver.h"
#include "tfile.h"
#include <cstring>
#include <memory>

using namespace optamd;

struct Opt_State {
    Opt_InitializationParameters params;
};
struct Opt_Problem {
    std::string filename, kind;
};
struct Opt_Plan {
    std::unique_ptr<SolverBase> solver;
    std::string energy;
    std::vector<std::string> kernelNames;
};

namespace optamd {
EnergyInfo imageWarpingInfo();
EnergyInfo poissonInfo();
EnergyInfo laplacianInfo();
EnergyInfo curveFittingInfo();
EnergyInfo arapInfo();
EnergyInfo sfsInfo();
EnergyInfo opticalFlowInfo();
EnergyInfo intrinsicInfo();
EnergyInfo volumetricInfo();
EnergyInfo cotangentInfo();
EnergyInfo embeddedInfo();
EnergyInfo robustInfo();
const std::vector<EnergyInfo>& energyRegistry() {
    static std::vector<EnergyInfo> reg = {imageWarpingInfo(), poissonInfo(), laplacianInfo(), curveFittingInfo(), arapInfo(), sfsInfo(),
                                           opticalFlowInfo(), intrinsicInfo(), volumetricInfo(),
                                           cotangentInfo(), embeddedInfo(), robustInfo()};
    return reg;
}
}  // namespace optamd

namespace {

const EnergyInfo* findEnergy(const std::string& stem) {
    for (auto& e : energyRegistry()) if (stem == e.name) return &e;
    return nullptr;
}

// Check the literal declarations of the .t against the registered binding layout.
bool validate(const TFile& tf, const EnergyInfo& info, std::string& why) {
    auto find = [&](const std::string& n) -> const ParamDecl* { for (auto& p : info.params) if (n == p.name) return &p; return nullptr; };
    std::vector<bool> seen(info.params.size(), false);
    auto mark = [&](const ParamDecl* p) { seen[p - &info.params[0]] = true; };
    for (auto& d : tf.decls) {
        if (d.kind == TDecl::kDim) {
            if (d.index >= info.nDims) { why = "Dim(\"" + d.name + "\"," + std::to_string(d.index) + ") exceeds the " + std::to_string(info.nDims) + " dimensions of the registered kernel set"; return false; }
            continue;
        }
        if (d.kind == TDecl::kGraph) {
            const ParamDecl* c = nullptr;
            for (auto& p : info.params) if (p.kind == ParamDecl::kGraphCount && d.name == p.name) c = &p;
            if (!c || c->index != d.index) { why = "Graph \"" + d.name + "\" does not match the registered binding layout"; return false; }
            mark(c);
            for (auto& s : d.slots) {
                const ParamDecl* p = find(d.name + "." + s.name);
                if (!p || p->kind != ParamDecl::kGraphIndex || p->index != s.index) { why = "Graph slot \"" + s.name + "\" does not match the registered binding layout"; return false; }
                mark(p);
            }
            continue;
        }
        const ParamDecl* p = find(d.name);
        if (!p) { why = "declaration \"" + d.name + "\" is not part of the registered kernel set"; return false; }
        const bool kindOk = (d.kind == TDecl::kUnknown && p->kind == ParamDecl::kUnknown) || (d.kind == TDecl::kArray && p->kind == ParamDecl::kArray) ||
                            (d.kind == TDecl::kParam && p->kind == ParamDecl::kScalar);
        std::string type = d.type.empty() ? "opt_float" : d.type;
        if (!kindOk || p->index != d.index || type != p->type) {
            why = "declaration \"" + d.name + "\" (type " + type + ", index " + std::to_string(d.index) + ") differs from the registered binding (type " + p->type + ", index " + std::to_string(p->index) + ")";
            return false;
        }
        mark(p);
    }
    for (size_t i = 0; i < info.params.size(); ++i)
        if (!seen[i] && info.params[i].kind != ParamDecl::kScalar) { why = std::string("the .t does not declare \"") + info.params[i].name + "\""; return false; }
    const bool usePre = tf.hasUsePreconditioner ? tf.usePreconditioner : false;   // default o.t:214
    if (usePre != info.usePreconditioner) { why = "UsePreconditioner differs from the registered kernel set"; return false; }
    return true;
}

}  // namespace

extern "C" {

Opt_State* Opt_NewState(Opt_InitializationParameters params) {
    if (params.threadsPerBlock <= 0 || params.threadsPerBlock % 32 != 0) params.threadsPerBlock = 256;   // createwrapper.t:148-151
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        // The product path is HIP-only: no CPU fallback exists.
        fprintf(stderr, "Opt_NewState: no HIP device available (%s); this backend has no CPU path\n", hipGetErrorString(e));
        return nullptr;
    }
    auto* s = new Opt_State{params};
    if (params.verbosityLevel > 1) {
        hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev);
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) printf("Opt (MI355X-native backend): device %d = %s (%s), %d CUs\n", dev, prop.name, prop.gcnArchName, prop.multiProcessorCount);
    }
    return s;
}

Opt_Problem* Opt_ProblemDefine(Opt_State* state, const char* filename, const char* solverkind) {
    if (!state || !filename || !solverkind) return nullptr;
    return new Opt_Problem{filename, solverkind};   // only recorded, like o.t:2521-2525
}
void Opt_ProblemDelete(Opt_State*, Opt_Problem* problem) { delete problem; }

Opt_Plan* Opt_ProblemPlan(Opt_State* state, Opt_Problem* problem, unsigned int* dimensions) {
    if (!state || !problem || !dimensions) return nullptr;
    // o.t:122: the solver kind must be one of the two known names
    // (plus this backend's extension "patchGaussNewtonGPU", OptAmd.h: block-local LDS-resident PCG sweeps, for energies that ship that kernel)
    const bool patch = problem->kind == "patchGaussNewtonGPU";
    if (problem->kind != "gaussNewtonGPU" && problem->kind != "LMGPU" && !patch) {
        fprintf(stderr, "Opt_ProblemPlan: expected solver kind to be gaussNewtonGPU or LMGPU, got '%s'\n", problem->kind.c_str());
        return nullptr;
    }
    const bool lm = !patch && problem->kind.find("LM") != std::string::npos;   // o.t:315
    TFile tf; std::string err;
    if (!readTFile(problem->filename, tf, err)) { fprintf(stderr, "Opt_ProblemPlan: %s\n", err.c_str()); return nullptr; }
    const EnergyInfo* info = findEnergy(tf.stem);
    if (!info) {
        fprintf(stderr, "Opt_ProblemPlan: energy '%s' (%s) has no hand-written kernel set in this backend. Registered:", tf.stem.c_str(), problem->filename.c_str());
        for (auto& e : energyRegistry()) fprintf(stderr, " %s", e.name);
        fprintf(stderr, "\n");
        return nullptr;
    }
    std::string why;
    if (!validate(tf, *info, why)) { fprintf(stderr, "Opt_ProblemPlan: %s: %s\n", problem->filename.c_str(), why.c_str()); return nullptr; }
    if (state->params.doublePrecision && info->floatOnly && state->params.verbosityLevel > 0)
        printf("Opt_ProblemPlan: '%s' declares fixed float unknowns; solving in float\n", info->name);
    SolverBase* s = makeSolver(*info, lm, patch, state->params.doublePrecision != 0, dimensions, state->params.collectPerKernelTimingInfo != 0, state->params.verbosityLevel);
    if (!s) { fprintf(stderr, "Opt_ProblemPlan: could not instantiate kernel set '%s'%s\n", info->name, patch ? " (it has no patch solver)" : ""); return nullptr; }
    if (state->params.verbosityLevel > 1) printf("Opt_ProblemPlan: %s (%s), nUnknowns = %ld, .t hash %016lx\n", info->name, problem->kind.c_str(), s->numUnknownScalars(), tf.bodyHash);
    auto* plan = new Opt_Plan; plan->solver.reset(s); plan->energy = info->name;
    return plan;
}
void Opt_PlanFree(Opt_State*, Opt_Plan* plan) { delete plan; }

void Opt_SetSolverParameter(Opt_State* state, Opt_Plan* plan, const char* name, void* value) {
    if (!plan || !name || !value) return;
    if (!plan->solver->setParameter(name, value) && state && state->params.verbosityLevel > 0)
        printf("Warning: tried to set nonexistent solver parameter %s\n", name);   // solver.t:1220
}
void Opt_ProblemInit(Opt_State*, Opt_Plan* plan, void** problemparams) { plan->solver->init(problemparams); }
int Opt_ProblemStep(Opt_State*, Opt_Plan* plan, void** problemparams) { return plan->solver->step(problemparams); }
void Opt_ProblemSolve(Opt_State* s, Opt_Plan* plan, void** problemparams) {   // o.t:2548-2551
    Opt_ProblemInit(s, plan, problemparams);
    while (Opt_ProblemStep(s, plan, problemparams)) {}
}
double Opt_ProblemCurrentCost(Opt_State*, Opt_Plan* plan) { return plan->solver->cost(); }

// ---- OptAmd.h ---------------------------------------------------------------------------------------------
const char* OptAmd_Version(void) { return "opt-amd 0.1 (gfx950, HIP; Opt API 0.2.2 compatible)"; }
int OptAmd_EnergyCount(void) { return (int)energyRegistry().size(); }
const char* OptAmd_EnergyName(int i) { return (i >= 0 && i < (int)energyRegistry().size()) ? energyRegistry()[i].name : nullptr; }
long OptAmd_PlanNumUnknownScalars(Opt_Plan* plan) { return plan->solver->numUnknownScalars(); }
void* OptAmd_PlanVector(Opt_Plan* plan, const char* name) { return plan->solver->vector(name); }
void OptAmd_EvalJTF(Opt_State*, Opt_Plan* plan, void** pp, void* jtf, void* diag) { plan->solver->evalJTF(pp, jtf, diag); }
double OptAmd_ApplyJTJ(Opt_State*, Opt_Plan* plan, void** pp, const void* v, void* out) { return plan->solver->applyJTJ(pp, v, out); }
double OptAmd_EvalCost(Opt_State*, Opt_Plan* plan, void** pp) { return plan->solver->evalCost(pp); }
void OptAmd_PlanEnableTrace(Opt_Plan* plan, int enable) { plan->solver->traceEnabled = enable != 0; }
long OptAmd_PlanTraceRows(Opt_Plan* plan) { return (long)plan->solver->trace.size() / 6; }
void OptAmd_PlanGetTrace(Opt_Plan* plan, double* rows6) { memcpy(rows6, plan->solver->trace.data(), plan->solver->trace.size() * sizeof(double)); }
double OptAmd_PlanTrustRegionRadius(Opt_Plan* plan) { return plan->solver->trustRegionRadius(); }
int OptAmd_PlanKernelTiming(Opt_Plan* plan, const char* kernel, long* count, double* total_ms) {
    auto& t = plan->solver->timer; t.evaluate();
    auto it = t.totals.find(kernel);
    if (it == t.totals.end()) return 0;
    if (count) *count = it->second.first; if (total_ms) *total_ms = it->second.second;
    return 1;
}
int OptAmd_PlanKernelCount(Opt_Plan* plan) { auto& t = plan->solver->timer; t.evaluate(); plan->kernelNames = t.order; return (int)plan->kernelNames.size(); }
const char* OptAmd_PlanKernelName(Opt_Plan* plan, int i) { return (i >= 0 && i < (int)plan->kernelNames.size()) ? plan->kernelNames[i].c_str() : nullptr; }
int OptAmd_CheckProblemFile(const char* filename, char* message, int messageLen) {
    TFile tf; std::string why;
    int ok = 0;
    if (!readTFile(filename, tf, why)) ok = 0;
    else {
        const EnergyInfo* info = findEnergy(tf.stem);
        if (!info) why = "energy '" + tf.stem + "' has no hand-written kernel set in this backend";
        else if (validate(tf, *info, why)) { ok = 1; why = std::string("ok: ") + info->name; }
    }
    if (message && messageLen > 0) { strncpy(message, why.c_str(), messageLen - 1); message[messageLen - 1] = 0; }
    return ok;
}
int OptAmd_PlanSetSlab(Opt_Plan* plan, long row0, long rows, long globalHeight, const OptAmd_SlabComm* comm) { return plan->solver->setSlab(row0, rows, globalHeight, comm); }

}  // extern "C"
