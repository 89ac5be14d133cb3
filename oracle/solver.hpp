// TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into, imported by, or called from the product path.
//
// CPU restatement of niessner/Opt's Gauss-Newton / Levenberg-Marquardt solver with the matrix-free
// PCG inner loop.  Sequencing, guards and defaults follow the reference line by line:
//   solver parameters / defaults ......... API/src/solverGPUGaussNewton.t:26-39, 148-163
//   PlanData vectors & scalars ........... solverGPUGaussNewton.t:167-214, 1254-1284
//   guardedInvert (CERES) ................ solverGPUGaussNewton.t:323-332
//   PCGInit1 / _Graph / _Finish .......... solverGPUGaussNewton.t:361-419, 687-692
//   PCGStep1 / _Graph .................... solverGPUGaussNewton.t:421-434, 694-706
//   PCGStep2 / 1stHalf / 2ndHalf ......... solverGPUGaussNewton.t:446-534
//   PCGStep3, LinearUpdate, revert, save . solverGPUGaussNewton.t:537-578
//   computeCost / computeModelCost ....... solverGPUGaussNewton.t:580-592, 666-678, 715-725, 746-756, 790-806
//   LM: CtC, SaveSSq, FinalizeDiagonal ... solverGPUGaussNewton.t:616-664, 739-744
//   init / step / cost ................... solverGPUGaussNewton.t:956-1007, 1016-1182
// What the generated per-energy functions mean (here evaluated generically from per-residual
// values + partials, the same information Opt's generator starts from):
//   cost = 1/2 sum r^2 ................... API/src/o.t:2375-2385
//   evalJTF (gradient w/o factor 2, diag)  o.t:2129-2172 (centred), 2228-2253 (graph)
//   applyJTJ ............................. o.t:2029-2089 (centred, + CtC*P for LM :2076-2082), 2092-2126 (graph)
//   computeCtC = diag(JtJ)/radius ........ o.t:2255-2316
//   modelcost = 1/2 sum (F + J delta)^2 .. o.t:2174-2225
//   exclude .............................. o.t:2452-2455 ; residual zeroed outside its bbox: o.t:1895-1936
//
// Reductions: the reference sums opt_float partials with warp shuffles + same-address atomics in an
// unspecified order (util.t:612-623, solver.t:312-317).  Two modes:
//   reductionMode 0 (default): every global sum accumulated in long double and rounded once to opt_float -- the "some order, no
//       accumulated error" limit;
//   reductionMode 1 ("reference order"): the reference's own arithmetic -- one opt_float term per element of the index space
//       (`p(idx):dot(Ap)`, `z:dot(r)`, `fmap.cost(idx)`), summed inside every 32-lane warp by the shfl.down tree of util.t:612-623 (a warp is
//       32 consecutive threads of a 16 x 16 block, x fastest: a 16 x 2 pixel patch; util.t:781-797), then ONE opt_float atomicAdd per warp
//       on a zeroed scalar (solverGPUGaussNewton.t:312-317, 580-592) in an order the hardware does not define -- modelled as a seeded random
//       permutation of the warps.  Two seeds are two legal runs of the reference; how far apart they land after N PCG iterations is the
//       reproducibility of the reference against ITSELF (tests/golden/make_reference_order_spread.py, profiles/r04_reference_order_spread.md).
//       Graph energies: the reference's graph kernels scatter J^T F and J^T J p with one opt_float atomicAdd per (vertex, channel) of every hyperedge
//       (o.t:2092-2126, 2228-2253; util.t:528-597) -- again in no defined order; in this mode the hyperedges are visited in a seeded random permutation (a new
//       one for every pass), the global sums stay exact.
//
// PARITY STATUS: the reference path is Terra/Lua JIT-compiled to PTX and cannot be built or run in this
// environment, and the reference ships no golden vectors.  This restatement is pinned by (i) the
// reference's one known-answer test (tests/minimal_graph_only: curve fit -> (100,102)), and (ii)
// finite-difference / symmetry identities (tests/test_oracle_*.py).  Beyond that: parity unpinned.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

namespace oracle {

constexpr int MAXS = 12;   // largest residual support (cotangent_mesh_smoothing edge: X of 4 vertices x 3)
constexpr int MAXR = 24;   // most scalar residuals per element (volumetric_mesh_deformation: 3 fit + 6 dirs x 3)

// One scalar residual instance: value, and partials w.r.t. the unknown scalars of its support.
// idx[k] = flat index into the unknown vector, or -1 if that access is outside the image
// (loads there return 0, o.t:570-576, and nothing gathers the partial).
template <class T>
struct Inst {
    int n;
    long idx[MAXS];
    T val;
    T dv[MAXS];
};

// Per-energy description: the residual templates of one .t file.
template <class T>
struct Energy {
    virtual ~Energy() {}
    bool usePreconditioner = false;   // o.t:214 default; set by UsePreconditioner() in the .t
    bool usesGraph = false;
    // unknown images in declaration order (iteration order of the unknown vector, o.t:675-687)
    std::vector<long> unkElems, unkChannels, unkOffset;
    long nScalars = 0;
    void addUnknown(long elems, long channels) {
        unkOffset.push_back(nScalars); unkElems.push_back(elems); unkChannels.push_back(channels);
        nScalars += elems * channels;
    }
    virtual void bind(void** params) = 0;                 // util.t:664-692: pointers & host scalars, re-read every init/step
    virtual T* unknownPtr(int img) = 0;                   // caller's array for unknown image img (updated in place)
    virtual long nCentered() const = 0;                   // elements of the index space carrying centred residuals
    virtual int evalCentered(long e, Inst<T>* out) const = 0;
    virtual bool excluded(int /*img*/, long /*elem*/) const { return false; }
    virtual bool excludedCentered(long /*elem*/) const { return false; }
    virtual long nEdges() const { return 0; }
    virtual int evalEdge(long /*e*/, Inst<T>* /*out*/) const { return 0; }
    virtual void precompute() {}                          // ComputedArrays (o.t:2387-2409)
    virtual long rowWidth() const { return 0; }            // image energies: W (lets the timed baseline split rows over threads)
};

struct SolverParameters {   // solver.t:148-163 (floats even in double mode), defaults :26-39
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
};

struct TraceRow { int nIter, lIter; double aNum, aDen, bNum, q; };

template <class T>
struct Solver {
    Energy<T>* E;
    bool lm;
    SolverParameters sp;
    // LM state lives in pd.parameters as opt_float (o.t:933-938, solver.t:998-1001)
    T trust_region_radius = 0, radius_decrease_factor = 0, min_lm_diagonal = 0, max_lm_diagonal = 0;
    T prevCost = 0;
    std::vector<T> delta, r, b, Adelta, z, p, Ap_X, CtC, preconditioner, SSq, prevX;
    std::vector<char> active;
    T aNum = 0, aDen = 0, bNum = 0, q = 0;
    std::vector<TraceRow> trace;
    std::vector<double> costHistory;
    int verbosity = 0;
    mutable std::vector<T> scratchAcc;
    int threads = 1;   // > 1: OpenMP over row bands (timed CPU baseline only; parity tests run single-threaded)
    int reductionMode = 0; unsigned reductionSeed = 0; mutable unsigned long reductionCount = 0;   // see the header: 1 = the reference's warp tree + unordered float atomics

    Solver(Energy<T>* e, bool useLM) : E(e), lm(useLM) {
        long n = E->nScalars;
        for (auto* v : {&delta, &r, &b, &Adelta, &z, &p, &Ap_X, &CtC, &preconditioner, &SSq, &prevX}) v->assign(n, T(0));
        active.assign(n, 1);
    }

    // ---- helpers ------------------------------------------------------------------------------
    void refreshActive() {
        for (size_t img = 0; img < E->unkElems.size(); ++img)
            for (long e = 0; e < E->unkElems[img]; ++e) {
                char a = E->excluded((int)img, e) ? 0 : 1;
                for (long c = 0; c < E->unkChannels[img]; ++c) active[E->unkOffset[img] + e * E->unkChannels[img] + c] = a;
            }
    }
    static T guardedInvert(T x) { T s = T(1) + std::sqrt(x); return T(1) / (s * s); }   // solver.t:323-332

    // ---- reductionMode 1: warp tree (util.t:612-623) + one opt_float atomicAdd per warp in a seeded random order (solver.t:312-317) ----
    bool referenceOrder() const { return reductionMode == 1 && !E->usesGraph && E->rowWidth() > 0; }
    T reduceReferenceOrder(const std::vector<T>& perElem) const {
        const long W = E->rowWidth(), n = (long)perElem.size(), H = n / W;
        const long bxN = (W + 15) / 16, byN = (H + 15) / 16;
        std::vector<T> warps((size_t)(bxN * byN * 8));
#pragma omp parallel for num_threads(threads) if (threads > 1)
        for (long b = 0; b < bxN * byN; ++b) {
            const long bx = b % bxN, by = b / bxN;
            for (int w = 0; w < 8; ++w) {
                T v[32];
                for (int lane = 0; lane < 32; ++lane) {      // thread (tx, ty) of the block, linear id = ty * 16 + tx; out-of-range threads contribute 0 (PCGStep1's `var d = 0`)
                    const long x = bx * 16 + (lane & 15), y = by * 16 + w * 2 + (lane >> 4);
                    v[lane] = (x < W && y < H) ? perElem[(size_t)(y * W + x)] : T(0);
                }
                for (int off = 16; off > 0; off >>= 1)       // val = val + __shfl_down(val, offset): lane 0's dependency cone
                    for (int lane = 0; lane < off; ++lane) v[lane] = v[lane] + v[lane + off];
                warps[(size_t)(b * 8 + w)] = v[0];
            }
        }
        // the atomics: a random permutation of the warps (Fisher-Yates on a 64-bit LCG stream keyed by the seed and the reduction's ordinal)
        unsigned long long st = 0x9E3779B97F4A7C15ull * (reductionSeed + 1) + 0xD1B54A32D192ED03ull * (++reductionCount);
        auto next = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned long long)(st >> 33); };
        for (size_t i = warps.size(); i > 1; --i) { const size_t j = (size_t)(next() % i); std::swap(warps[i - 1], warps[j]); }
        T s = 0;
        for (T x : warps) s = s + x;
        return s;
    }
    // one opt_float per element: term(e), evaluated in parallel, then reduced as above
    template <class F> T reducePerElement(F&& term) const {
        const long n = E->nCentered();
        std::vector<T> perElem((size_t)n);
#pragma omp parallel for num_threads(threads) if (threads > 1)
        for (long e = 0; e < n; ++e) perElem[(size_t)e] = term(e);
        return reduceReferenceOrder(perElem);
    }
    // unknownElement dot product at element e (all unknown images live on the same index space): sum over images, channels of a * c
    T elemDot(const std::vector<T>& a, const std::vector<T>& c, long e) const {
        T s = 0;
        for (size_t img = 0; img < E->unkElems.size(); ++img) {
            const long ch = E->unkChannels[img], base = E->unkOffset[img] + e * ch;
            if (!active[base]) return T(0);
            for (long k = 0; k < ch; ++k) s = s + a[base + k] * c[base + k];
        }
        return s;
    }

    template <class F> void forEachInstance(bool skipExcludedCentres, F&& f) const {
        Inst<T> buf[MAXR];
        long nc = E->nCentered();
        for (long e = 0; e < nc; ++e) {
            if (skipExcludedCentres && E->excludedCentered(e)) continue;
            int k = E->evalCentered(e, buf);
            f(buf, k);
        }
        long ne = E->nEdges();
        if (reductionMode == 1 && E->usesGraph && ne > 1) {      // the scatter order of the reference's per-edge atomics is undefined: a seeded permutation per pass
            edgeOrder(ne);
            for (long j = 0; j < ne; ++j) { int k = E->evalEdge(edgePerm[(size_t)j], buf); f(buf, k); }
            return;
        }
        for (long e = 0; e < ne; ++e) {
            int k = E->evalEdge(e, buf);
            f(buf, k);
        }
    }
    mutable std::vector<long> edgePerm;
    void edgeOrder(long ne) const {
        edgePerm.resize((size_t)ne);
        for (long e = 0; e < ne; ++e) edgePerm[(size_t)e] = e;
        unsigned long long st = 0x9E3779B97F4A7C15ull * (reductionSeed + 1) + 0xD1B54A32D192ED03ull * (++reductionCount);
        auto next = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned long long)(st >> 33); };
        for (size_t i = edgePerm.size(); i > 1; --i) { const size_t j = (size_t)(next() % i); std::swap(edgePerm[i - 1], edgePerm[j]); }
    }

    // Row-banded traversal of the centred instances for the multi-threaded baseline.  Every residual's support lies
    // within +-1 row of its centre, so bands of 4 rows processed in two colours (even bands, then odd bands) never
    // scatter into the same row from two threads; per-band partial sums keep reductions deterministic.
    template <class F> void forEachInstanceBanded(bool skipExcludedCentres, F&& f, std::vector<long double>* bandSums = nullptr) const {
        const long W = E->rowWidth(), nc = E->nCentered();
        if (threads <= 1 || W <= 0 || nc % W != 0) {
            long double s = 0; long double* sp = bandSums ? &s : nullptr;
            forEachInstance(skipExcludedCentres, [&](const Inst<T>* in, int k) { f(in, k, sp); });
            if (bandSums) bandSums->assign(1, s);
            return;
        }
        const long H = nc / W, band = 4, nb = (H + band - 1) / band;
        if (bandSums) bandSums->assign(nb + 1, 0.0L);
        for (int colour = 0; colour < 2; ++colour) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
            for (long bi = colour; bi < nb; bi += 2) {
                Inst<T> buf[MAXR];
                long double s = 0;
                const long e0 = bi * band * W, e1 = std::min(nc, (bi + 1) * band * W);
                for (long e = e0; e < e1; ++e) {
                    if (skipExcludedCentres && E->excludedCentered(e)) continue;
                    int k = E->evalCentered(e, buf);
                    f(buf, k, bandSums ? &s : nullptr);
                }
                if (bandSums) (*bandSums)[bi] = s;
            }
        }
        Inst<T> buf[MAXR];
        long double s = 0;
        const long ne = E->nEdges();
        const bool perm = reductionMode == 1 && E->usesGraph && ne > 1;
        if (perm) edgeOrder(ne);
        for (long j = 0; j < ne; ++j) { int k = E->evalEdge(perm ? edgePerm[(size_t)j] : j, buf); f(buf, k, bandSums ? &s : nullptr); }
        if (bandSums) (*bandSums)[nb] = s;
    }
    static long double sumBands(const std::vector<long double>& v) { long double s = 0; for (auto x : v) s += x; return s; }

    // cost = sum over non-excluded elements of 1/2 sum_k r_k^2  (solver.t:580-592, 715-725; o.t:2375-2385)
    T computeCost() const {
        if (referenceOrder())      // computeCost kernel, solver.t:580-592: cost = fmap.cost(idx), warpReduce, lane 0 atomicAdd
            return reducePerElement([&](long e) {
                if (E->excludedCentered(e)) return T(0);
                Inst<T> buf[MAXR];
                const int k = E->evalCentered(e, buf);
                T c = 0; for (int i = 0; i < k; ++i) c += buf[i].val * buf[i].val;
                return T(0.5) * c;
            });
        std::vector<long double> bands;
        forEachInstanceBanded(true, [&](const Inst<T>* in, int k, long double* s) {
            T c = 0; for (int i = 0; i < k; ++i) c += in[i].val * in[i].val;
            *s += (long double)(T(0.5) * c);
        }, &bands);
        return (T)sumBands(bands);
    }
    // modelcost = 1/2 sum (F + J delta)^2  (o.t:2174-2225)
    T computeModelCost() const {
        std::vector<long double> bands;
        forEachInstanceBanded(true, [&](const Inst<T>* in, int k, long double* s) {
            T c = 0;
            for (int i = 0; i < k; ++i) {
                T jd = 0;
                for (int u = 0; u < in[i].n; ++u) if (in[i].idx[u] >= 0) jd += in[i].dv[u] * delta[in[i].idx[u]];
                T m = in[i].val + jd; c += m * m;
            }
            *s += (long double)(T(0.5) * c);
        }, &bands);
        return (T)sumBands(bands);
    }
    // F^ = sum dr/dx * r ; P^ = sum (dr/dx)^2 over every residual touching x (incl. residuals centred on
    // excluded neighbours: the gather of o.t:2045-2064 has no exclude test).
    void evalJTF(std::vector<T>& F, std::vector<T>& P) const {
        F.assign(E->nScalars, T(0)); P.assign(E->nScalars, T(0));
        forEachInstanceBanded(false, [&](const Inst<T>* in, int k, long double*) {
            for (int i = 0; i < k; ++i)
                for (int u = 0; u < in[i].n; ++u) {
                    long t = in[i].idx[u];
                    if (t < 0) continue;
                    F[t] += in[i].dv[u] * in[i].val;
                    P[t] += in[i].dv[u] * in[i].dv[u];
                }
        });
    }
    // out = J^T J v (+ CtC .* v for LM, o.t:2076-2082); only active rows are produced (solver.t:424).
    void applyJTJ(const std::vector<T>& v, std::vector<T>& out) const {
        if ((long)scratchAcc.size() != E->nScalars) scratchAcc.resize(E->nScalars);
        std::vector<T>& acc = scratchAcc;
#pragma omp parallel for num_threads(threads) if (threads > 1)
        for (long i = 0; i < E->nScalars; ++i) acc[i] = T(0);
        forEachInstanceBanded(false, [&](const Inst<T>* in, int k, long double*) {
            for (int i = 0; i < k; ++i) {
                T jp = 0;
                for (int u = 0; u < in[i].n; ++u) if (in[i].idx[u] >= 0) jp += in[i].dv[u] * v[in[i].idx[u]];
                for (int u = 0; u < in[i].n; ++u) if (in[i].idx[u] >= 0) acc[in[i].idx[u]] += in[i].dv[u] * jp;
            }
        });
#pragma omp parallel for num_threads(threads) if (threads > 1)
        for (long i = 0; i < E->nScalars; ++i) {
            if (!active[i]) continue;
            out[i] = acc[i] + (lm ? CtC[i] * v[i] : T(0));
        }
    }
    T dotActive(const std::vector<T>& a, const std::vector<T>& c) const {
        if (referenceOrder()) return reducePerElement([&](long e) { return elemDot(a, c, e); });      // d = pd.p(idx):dot(tmp), solver.t:429
        long double s = 0;
#pragma omp parallel for reduction(+ : s) num_threads(threads) if (threads > 1)
        for (long i = 0; i < E->nScalars; ++i) if (active[i]) s += (long double)(a[i] * c[i]);
        return (T)s;
    }
    void linearUpdate(T sign) {   // X += delta  (solver.t:552-557)
        for (size_t img = 0; img < E->unkElems.size(); ++img) {
            T* X = E->unknownPtr((int)img);
            long n = E->unkElems[img] * E->unkChannels[img], off = E->unkOffset[img];
            for (long i = 0; i < n; ++i) if (active[off + i]) X[i] += sign * delta[off + i];
        }
    }
    void saveOrRevert(bool save) {   // solver.t:559-564, 573-578
        for (size_t img = 0; img < E->unkElems.size(); ++img) {
            T* X = E->unknownPtr((int)img);
            long n = E->unkElems[img] * E->unkChannels[img], off = E->unkOffset[img];
            for (long i = 0; i < n; ++i) if (active[off + i]) { if (save) prevX[off + i] = X[i]; else X[i] = prevX[off + i]; }
        }
    }

    // ---- init (solver.t:956-1007) -------------------------------------------------------------
    void init(void** params) {
        E->bind(params);
        refreshActive();
        sp.nIter = 0;
        if (lm) {
            trust_region_radius = (T)sp.trust_region_radius;
            radius_decrease_factor = (T)sp.radius_decrease_factor;
            min_lm_diagonal = (T)sp.min_lm_diagonal;
            max_lm_diagonal = (T)sp.max_lm_diagonal;
        }
        E->precompute();
        prevCost = computeCost();
        trace.clear(); costHistory.clear();
        costHistory.push_back((double)prevCost);
    }

    // ---- the PCG pieces -----------------------------------------------------------------------
    void pcgInit1() {   // solver.t:361-419 + 687-692
        std::vector<T> F, P;
        evalJTF(F, P);
        long double d = 0;
        const long n = E->nScalars;
        if (!E->usesGraph) {
            for (long i = 0; i < n; ++i) {
                T residuum = 0, pre = 0;
                if (active[i]) {
                    delta[i] = 0;
                    residuum = -F[i];
                    r[i] = residuum;
                    pre = E->usePreconditioner ? P[i] : T(1);
                    pre = guardedInvert(pre);
                    T pp = pre * residuum;
                    p[i] = pp;
                    d += (long double)(residuum * pp);
                }
                preconditioner[i] = pre;
            }
        } else {
            // PCGInit1 (centred part) writes r = -F^c and pre = P^c (the constant 1 when not preconditioning,
            // o.t:2163-2164), PCGInit1_Graph atomically adds -JtF and J^2 on top (evalJTF above already holds
            // centred + graph sums), PCGInit1_Finish inverts and -- when not preconditioning -- overrides to 1.
            for (long i = 0; i < n; ++i) {
                if (!active[i]) { preconditioner[i] = 0; continue; }
                delta[i] = 0;
                r[i] = -F[i];
                // PCGInit1_Finish (solver.t:399-419)
                T pre = E->usePreconditioner ? guardedInvert(P[i]) : T(1);
                T pp = pre * r[i];
                preconditioner[i] = pre;
                p[i] = pp;
                d += (long double)(r[i] * pp);
            }
        }
        aNum = (T)d;
        if (referenceOrder()) aNum = reducePerElement([&](long e) { return elemDot(r, p, e); });      // d = residuum:dot(p), solver.t:391
    }
    void computeCtC() {   // solver.t:616-622, 739-744 ; o.t:2255-2316 (true diag(JtJ)/radius, independent of usepreconditioner)
        std::vector<T> acc(E->nScalars, T(0));
        T inv_radius = T(1) / trust_region_radius;
        forEachInstanceBanded(false, [&](const Inst<T>* in, int k, long double*) {
            for (int i = 0; i < k; ++i) for (int u = 0; u < in[i].n; ++u) if (in[i].idx[u] >= 0) acc[in[i].idx[u]] += in[i].dv[u] * in[i].dv[u] * inv_radius;
        });
        for (long i = 0; i < E->nScalars; ++i) if (active[i]) CtC[i] = acc[i];
    }
    void finalizeDiagonal() {   // solver.t:631-664
        long double d = 0, qq = 0;
        for (long i = 0; i < E->nScalars; ++i) {
            if (!active[i]) continue;
            T unclamped = CtC[i];
            T invS = T(1) / SSq[i];
            T clampMul = invS / trust_region_radius;
            T minVal = min_lm_diagonal * clampMul, maxVal = max_lm_diagonal * clampMul;
            T c = std::fmin(std::fmax(unclamped, minVal), maxVal);
            CtC[i] = c;
            T pre = T(1) / (c + trust_region_radius * unclamped);
            preconditioner[i] = pre;
            T residuum = r[i];
            b[i] = residuum;
            T pp = pre * residuum;
            p[i] = pp;
            d += (long double)(residuum * pp);
            qq += (long double)(T(0.5) * (delta[i] * (residuum + residuum)));
        }
        q = (T)qq; aNum = (T)d;
    }
    void pcgStep1() {   // solver.t:421-434, 694-706
        applyJTJ(p, Ap_X);
        aDen = dotActive(p, Ap_X);
    }
    T alpha() const { return (aDen > T(0)) ? aNum / aDen : T(0); }   // guardDivisionByZero, solver.t:456-459
    void pcgStep2() {   // solver.t:446-489
        T a = alpha();
        long double bn = 0, qq = 0;
#pragma omp parallel for reduction(+ : bn, qq) num_threads(threads) if (threads > 1)
        for (long i = 0; i < E->nScalars; ++i) {
            if (!active[i]) continue;
            T dl = delta[i] + a * p[i]; delta[i] = dl;
            T rr = r[i] - a * Ap_X[i]; r[i] = rr;
            T pre = E->usePreconditioner ? preconditioner[i] : T(1);
            T zz = pre * rr; z[i] = zz;
            bn += (long double)(zz * rr);
            if (lm) qq += (long double)(T(0.5) * (dl * (rr + b[i])));
        }
        bNum = (T)bn; if (lm) q = (T)qq;
        if (referenceOrder()) {      // betaNum = z:dot(r); q = 0.5 * delta:dot(r + b)  (solver.t:478-485)
            bNum = reducePerElement([&](long e) { return elemDot(z, r, e); });
            if (lm) {
                std::vector<T> rb(r.size());
                for (size_t i = 0; i < r.size(); ++i) rb[i] = r[i] + b[i];
                q = reducePerElement([&](long e) { return T(0.5) * elemDot(delta, rb, e); });
            }
        }
    }
    void pcgStep2Split() {   // solver.t:491-534 + computeAdelta :566-571, 708-713
        T a = alpha();
        for (long i = 0; i < E->nScalars; ++i) if (active[i]) delta[i] = delta[i] + a * p[i];
        applyJTJ(delta, Adelta);
        long double bn = 0, qq = 0;
        for (long i = 0; i < E->nScalars; ++i) {
            if (!active[i]) continue;
            T rr = b[i] - Adelta[i]; r[i] = rr;
            T pre = E->usePreconditioner ? preconditioner[i] : T(1);
            T zz = pre * rr; z[i] = zz;
            bn += (long double)(zz * rr);
            qq += (long double)(T(0.5) * (delta[i] * (rr + b[i])));
        }
        bNum = (T)bn; q = (T)qq;
    }
    void pcgStep3() {   // solver.t:537-550
        T beta = (aNum > T(0)) ? bNum / aNum : T(0);
#pragma omp parallel for num_threads(threads) if (threads > 1)
        for (long i = 0; i < E->nScalars; ++i) if (active[i]) p[i] = z[i] + beta * p[i];
    }

    // ---- step (solver.t:1016-1177) ------------------------------------------------------------
    int step(void** params) {
        const T min_relative_decrease = (T)sp.min_relative_decrease;
        const T min_trust_region_radius = (T)sp.min_trust_region_radius;
        const T max_trust_region_radius = (T)sp.max_trust_region_radius;
        const T q_tolerance = (T)sp.q_tolerance;
        const T function_tolerance = (T)sp.function_tolerance;
        T Q0 = 0, Q1 = 0;
        E->bind(params);
        refreshActive();
        if (sp.nIter >= sp.nIterations) return 0;

        aNum = aDen = bNum = 0;
        pcgInit1();
        if (lm) {
            aNum = 0; q = 0;
            if (sp.nIter == 0) for (long i = 0; i < E->nScalars; ++i) if (active[i]) SSq[i] = preconditioner[i];   // PCGSaveSSq
            computeCtC();
            finalizeDiagonal();
            Q0 = q;
        }
        for (int lIter = 0; lIter < sp.lIterations; ++lIter) {
            aDen = 0; q = 0;
            pcgStep1();
            bNum = 0;
            if (lm && ((lIter + 1) % sp.residual_reset_period) == 0) pcgStep2Split();
            else pcgStep2();
            pcgStep3();
            trace.push_back({sp.nIter, lIter, (double)aNum, (double)aDen, (double)bNum, (double)q});
            aNum = bNum;   // solver.t:1091
            if (lm) {
                Q1 = q;
                T zeta = T(lIter + 1) * (Q1 - Q0) / Q1;
                if (zeta < q_tolerance) { if (verbosity) printf("zeta=%.18g, breaking at iteration: %d\n", (double)zeta, lIter + 1); break; }
                Q0 = Q1;
            }
        }
        T model_cost_change = 0;
        if (lm) {
            T model_cost = computeModelCost();
            model_cost_change = prevCost - model_cost;
            saveOrRevert(true);
        }
        linearUpdate(T(1));
        E->precompute();
        T newCost = computeCost();
        if (lm) {
            T cost_change = prevCost - newCost;
            T relative_decrease = cost_change / model_cost_change;
            if (cost_change >= 0 && relative_decrease > min_relative_decrease) {
                T absolute_function_tolerance = prevCost * function_tolerance;
                if (cost_change <= absolute_function_tolerance) { costHistory.push_back((double)prevCost); return 0; }
                // Terra promotes these literals to double (solver.t:1135-1139); results are stored back as opt_float.
                double step_quality = (double)relative_decrease;
                double min_factor = 1.0 / 3.0;
                double tmp_factor = 1.0 - std::pow(2.0 * step_quality - 1.0, 3.0);
                trust_region_radius = (T)((double)trust_region_radius / std::fmax(min_factor, tmp_factor));
                trust_region_radius = std::fmin(trust_region_radius, max_trust_region_radius);
                radius_decrease_factor = T(2.0);
                prevCost = newCost;
            } else {
                saveOrRevert(false);
                trust_region_radius = trust_region_radius / radius_decrease_factor;
                radius_decrease_factor = T(2.0) * radius_decrease_factor;
                if (trust_region_radius <= min_trust_region_radius) { costHistory.push_back((double)prevCost); return 0; }
                E->precompute();
            }
        } else {
            if (verbosity) printf("cost: %f -> %f\n", (double)prevCost, (double)newCost);
            prevCost = newCost;
        }
        costHistory.push_back((double)prevCost);
        sp.nIter += 1;
        return 1;
    }
};

}  // namespace oracle
