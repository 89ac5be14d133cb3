// TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into, imported by, or called from the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// C entry points of the CPU oracle.  The shape mirrors Opt.h (reference API/release/include/Opt.h:35-71)
// -- create / set parameter / init / step / cost / free -- but every pointer in `params` is a HOST
// pointer, and the energy is named directly instead of being read from a .t file.  Extra probes expose
// the intermediate vectors so tests can compare them with the HIP solver's.
#include "energies.hpp"
#include "sfs.hpp"
#include "energies_grid.hpp"
#include "energies_mesh.hpp"
#include "patch.hpp"
#include <memory>

using namespace oracle;

namespace {
struct Handle {
    bool dbl = false;
    std::unique_ptr<Energy<float>> ef; std::unique_ptr<Solver<float>> sf;
    std::unique_ptr<Energy<double>> ed; std::unique_ptr<Solver<double>> sd;
};
template <class T> Energy<T>* makeEnergy(const std::string& n, const unsigned* dims) {
    if (n == "image_warping") return new ImageWarping<T>(dims);
    if (n == "poisson_image_editing") return new Poisson<T>(dims);
    if (n == "laplacian") return new Laplacian<T>(dims);
    if (n == "curveFitting") return new CurveFitting<T>(dims);
    if (n == "arap_mesh_deformation") return new Arap<T>(dims);
    if (n == "shape_from_shading") return new ShapeFromShading<T>(dims);
    if (n == "optical_flow") return new OpticalFlow<T>(dims);
    if (n == "intrinsic_image_decomposition") return new IntrinsicImage<T>(dims);
    if (n == "volumetric_mesh_deformation") return new VolumetricMesh<T>(dims);
    if (n == "cotangent_mesh_smoothing") return new CotangentSmoothing<T>(dims);
    if (n == "embedded_mesh_deformation") return new EmbeddedDeformation<T>(dims);
    if (n == "robust_nonrigid_alignment") return new RobustAlignment<T>(dims);
    return nullptr;
}
template <class T> std::vector<T>* vecByName(Solver<T>* s, const std::string& n) {
    if (n == "delta") return &s->delta; if (n == "r") return &s->r; if (n == "b") return &s->b;
    if (n == "Adelta") return &s->Adelta; if (n == "z") return &s->z; if (n == "p") return &s->p;
    if (n == "Ap_X") return &s->Ap_X; if (n == "CtC") return &s->CtC; if (n == "preconditioner") return &s->preconditioner;
    if (n == "SSq") return &s->SSq; if (n == "prevX") return &s->prevX;
    return nullptr;
}
bool setParam(SolverParameters& sp, const char* name, const void* v) {   // solver.t:1205-1221
    std::string n(name);
#define F(x) if (n == #x) { sp.x = *(const float*)v; return true; }
#define I(x) if (n == #x) { sp.x = *(const int*)v; return true; }
    F(min_relative_decrease) F(min_trust_region_radius) F(max_trust_region_radius) F(q_tolerance) F(function_tolerance)
    F(trust_region_radius) F(radius_decrease_factor) F(min_lm_diagonal) F(max_lm_diagonal)
    I(residual_reset_period) I(nIter) I(nIterations) I(lIterations)
#undef F
#undef I
    return false;
}
}  // namespace

extern "C" {

void* OptOracle_Create(const char* energy, const char* solverkind, int doublePrecision, const unsigned* dims) {
    std::string kind(solverkind);
    if (kind != "gaussNewtonGPU" && kind != "LMGPU") return nullptr;   // o.t:122
    bool lm = kind.find("LM") != std::string::npos;                     // o.t:315
    auto* h = new Handle; h->dbl = doublePrecision != 0;
    if (h->dbl) { h->ed.reset(makeEnergy<double>(energy, dims)); if (!h->ed) { delete h; return nullptr; } h->sd.reset(new Solver<double>(h->ed.get(), lm)); }
    else { h->ef.reset(makeEnergy<float>(energy, dims)); if (!h->ef) { delete h; return nullptr; } h->sf.reset(new Solver<float>(h->ef.get(), lm)); }
    return h;
}
void OptOracle_Free(void* hv) { delete (Handle*)hv; }
int OptOracle_SetSolverParameter(void* hv, const char* name, const void* value) {
    auto* h = (Handle*)hv; return setParam(h->dbl ? h->sd->sp : h->sf->sp, name, value) ? 1 : 0;
}
void OptOracle_Init(void* hv, void** params) { auto* h = (Handle*)hv; if (h->dbl) h->sd->init(params); else h->sf->init(params); }
int OptOracle_Step(void* hv, void** params) { auto* h = (Handle*)hv; return h->dbl ? h->sd->step(params) : h->sf->step(params); }
void OptOracle_Solve(void* hv, void** params) { OptOracle_Init(hv, params); while (OptOracle_Step(hv, params)) {} }   // o.t:2548-2551
double OptOracle_CurrentCost(void* hv) { auto* h = (Handle*)hv; return h->dbl ? (double)h->sd->prevCost : (double)h->sf->prevCost; }
// reductionMode 1: the reference's warp tree + unordered opt_float atomics, the order drawn from `seed` (solver.hpp header)
void OptOracle_SetReduction(void* hv, int mode, unsigned seed) {
    auto* h = (Handle*)hv;
    if (h->dbl) { h->sd->reductionMode = mode; h->sd->reductionSeed = seed; h->sd->reductionCount = 0; }
    else { h->sf->reductionMode = mode; h->sf->reductionSeed = seed; h->sf->reductionCount = 0; }
}
// Process-wide: which float sin / cos the restatement evaluates (dual.hpp trigSeed: 0 = the host libm; n > 0 = a seeded implementation within 1 ulp)
void OptOracle_SetTrigVariant(unsigned seed) { oracle::trigSeed() = seed; }
void OptOracle_SetThreads(void* hv, int n) { auto* h = (Handle*)hv; if (h->dbl) h->sd->threads = n; else h->sf->threads = n; }
long OptOracle_NumUnknownScalars(void* hv) { auto* h = (Handle*)hv; return h->dbl ? h->ed->nScalars : h->ef->nScalars; }

// --- probes -------------------------------------------------------------------------------------
int OptOracle_GetVector(void* hv, const char* name, void* out) {
    auto* h = (Handle*)hv;
    if (h->dbl) { auto* v = vecByName(h->sd.get(), name); if (!v) return 0; memcpy(out, v->data(), v->size() * sizeof(double)); }
    else { auto* v = vecByName(h->sf.get(), name); if (!v) return 0; memcpy(out, v->data(), v->size() * sizeof(float)); }
    return 1;
}
// binds params, evaluates the raw gradient F^ = J^T F and diagonal P^ = diag(J^T J) at the current unknowns
void OptOracle_EvalJTF(void* hv, void** params, void* jtf, void* diag) {
    auto* h = (Handle*)hv;
    if (h->dbl) { auto* s = h->sd.get(); s->E->bind(params); s->refreshActive(); s->E->precompute(); std::vector<double> F, P; s->evalJTF(F, P);
        memcpy(jtf, F.data(), F.size() * 8); memcpy(diag, P.data(), P.size() * 8); }
    else { auto* s = h->sf.get(); s->E->bind(params); s->refreshActive(); s->E->precompute(); std::vector<float> F, P; s->evalJTF(F, P);
        memcpy(jtf, F.data(), F.size() * 4); memcpy(diag, P.data(), P.size() * 4); }
}
// out = J^T J v on active rows (0 elsewhere); CtC is NOT added (pure Gauss-Newton operator)
void OptOracle_ApplyJTJ(void* hv, void** params, const void* v, void* out) {
    auto* h = (Handle*)hv;
    if (h->dbl) { auto* s = h->sd.get(); s->E->bind(params); s->refreshActive(); s->E->precompute(); long n = s->E->nScalars;
        std::vector<double> vin((const double*)v, (const double*)v + n), o(n, 0.0); bool lm = s->lm; s->lm = false; s->applyJTJ(vin, o); s->lm = lm; memcpy(out, o.data(), n * 8); }
    else { auto* s = h->sf.get(); s->E->bind(params); s->refreshActive(); s->E->precompute(); long n = s->E->nScalars;
        std::vector<float> vin((const float*)v, (const float*)v + n), o(n, 0.f); bool lm = s->lm; s->lm = false; s->applyJTJ(vin, o); s->lm = lm; memcpy(out, o.data(), n * 4); }
}
double OptOracle_EvalCost(void* hv, void** params) {
    auto* h = (Handle*)hv;
    if (h->dbl) { auto* s = h->sd.get(); s->E->bind(params); s->refreshActive(); s->E->precompute(); return (double)s->computeCost(); }
    auto* s = h->sf.get(); s->E->bind(params); s->refreshActive(); s->E->precompute(); return (double)s->computeCost();
}
// per-PCG-iteration scalars: rows of {nIter, lIter, alphaNum, alphaDen, betaNum, q}
long OptOracle_TraceRows(void* hv) { auto* h = (Handle*)hv; return (long)(h->dbl ? h->sd->trace.size() : h->sf->trace.size()); }
void OptOracle_GetTrace(void* hv, double* out) {
    auto* h = (Handle*)hv; const auto& t = h->dbl ? h->sd->trace : h->sf->trace;
    for (size_t i = 0; i < t.size(); ++i) { out[6 * i] = t[i].nIter; out[6 * i + 1] = t[i].lIter; out[6 * i + 2] = t[i].aNum; out[6 * i + 3] = t[i].aDen; out[6 * i + 4] = t[i].bNum; out[6 * i + 5] = t[i].q; }
}
long OptOracle_CostHistoryLen(void* hv) { auto* h = (Handle*)hv; return (long)(h->dbl ? h->sd->costHistory.size() : h->sf->costHistory.size()); }
void OptOracle_GetCostHistory(void* hv, double* out) {
    auto* h = (Handle*)hv; const auto& c = h->dbl ? h->sd->costHistory : h->sf->costHistory;
    for (size_t i = 0; i < c.size(); ++i) out[i] = c[i];
}
// block-local patch solver for poisson_image_editing (patch.hpp): X in/out, costs[nIterations + 1]
void OptOracle_PoissonPatchSolve(int doublePrecision, int W, int H, void* X, const void* T, const void* M, int nIterations, int lIterations,
                                 int patchIterations, int patchSize, double* costs) {
    if (doublePrecision) { PoissonPatch<double> p{W, H, (const double*)T, (const double*)M}; p.solve((double*)X, nIterations, lIterations, patchIterations, patchSize, costs); }
    else { PoissonPatch<float> p{W, H, (const float*)T, (const float*)M}; p.solve((float*)X, nIterations, lIterations, patchIterations, patchSize, costs); }
}
double OptOracle_TrustRegionRadius(void* hv) { auto* h = (Handle*)hv; return h->dbl ? (double)h->sd->trust_region_radius : (double)h->sf->trust_region_radius; }

}  // extern "C"
