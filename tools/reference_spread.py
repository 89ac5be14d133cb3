"""One yardstick for long-horizon parity: how far apart do LEGAL runs of the reference's arithmetic end?

The reference sums its dot products with one opt_float atomicAdd per warp (API/src/util.t:612-623, API/src/solverGPUGaussNewton.t:312-317): the order in
which the N/32 atomics commit is not defined, so two runs of the reference itself differ.  The oracle's reference-order mode (oracle/solver.hpp: per-element
opt_float terms, the 32-lane shfl.down tree, the per-warp partials added in a seeded random order) reproduces that; every seed is one legal run.  The frozen
runs are in tests/golden/:

    horizon_costs.json / horizon_costs_fma.json       exact-order sums (long double), plain build / fused-multiply-add build of the oracle
    reference_order_costs.json                        reference-order sums, seeds 1..5 (plain build) and 1..3 (fma build, keys *_fma)

spread(key, step)  = the diameter of that set (largest pairwise relative distance of the cost after `step` Gauss-Newton steps);
yardstick          = max(contract floor, spread)            contract floor: 1e-5 float, 1e-12 double (BASELINE.json north_star);
a HIP loop is `within_reference_spread` if its cost is at most one yardstick from the exact-order plain oracle (FACTOR = 1 since round 5).

Round 5 adds a second, physical yardstick for the `horizon` family (profiles/r05_l50_bisect.md): the exact-order oracle is frozen in BOTH builds (plain and with fused
multiply-adds: two legal roundings of the same elementwise arithmetic; the HIP compiler contracts like the second) and at the horizons next to each tested one, so a
HIP cost can be placed relative to the HULL of the two exact-order runs in units of ONE PCG ITERATION OF PROGRESS, |c(L-1) - c(L+1)| / 2 of the plain oracle
(`iterations_from_hull`): the cost of this solve still falls by 0.5 % per iteration at L = 50, and its ~5.5-iteration cycle of near-breakdowns amplifies 1e-7
differences 1e5-fold for two iterations at a time -- a diameter of scalar-noise runs does not measure that, an iteration of progress does.

No GPU, no oracle library: this module only reads the frozen numbers (tests/test_horizon_gpu.py, tools/horizon_parity.py, bench.py).
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FLOOR = {"float": 1e-5, "double": 1e-12}
FACTOR = 1.0      # (round 5: no allowance on top of the yardstick; round 4 used 2)

# key suffixes of the reference-order runs: build (plain / fused multiply-adds) x accumulation order of the J^T J p scatter (banded two-colour traversal of the
# multi-threaded oracle / raster order of the single-threaded one) x float sin / cos (the host libm / `trig`: a seeded implementation within 1 ulp, oracle/dual.hpp --
# the reference calls libdevice's) -- every combination is the same algorithm under another legal rounding
VARIANTS = (("", "plain"), ("_fma", "fma"), ("_raster", "raster"), ("_raster_fma", "raster fma"), ("_trig", "trig"), ("_trig_fma", "trig fma"))
_cache = {}


def _load(name):
    if name not in _cache:
        p = os.path.join(GOLD, name)
        _cache[name] = json.load(open(p)) if os.path.exists(p) else {}
    return _cache[name]


def _reference_order():
    """reference_order_costs.json merged with any reference_order_costs_*.json beside it (more seeds of some workloads, generated separately)."""
    if "_ro" not in _cache:
        import glob
        R = {}
        for p in sorted(glob.glob(os.path.join(GOLD, "reference_order_costs*.json"))):
            for k, e in json.load(open(p)).items():
                R.setdefault(k, {"costs_by_seed": {}})["costs_by_seed"].update(e.get("costs_by_seed", {}))
        _cache["_ro"] = R
    return _cache["_ro"]


def _exact(key, name):
    """The exact-order run of `key` in golden file `name`.  bench_<size>_<precision>_400x2 (bench.py's first two Gauss-Newton steps) is frozen in bench_costs.json."""
    if key.startswith("bench_"):
        _, size, prec, _ = key.split("_")
        if name != "horizon_costs.json":      # the fma build's exact-order run of these two steps: the start of its frozen 8 x 400 solve (same workload, same initial guess)
            return _load(name).get(f"solve8_{size}_{prec}")
        return _load("bench_costs.json").get(f"image_warping_{size}x{size}_{prec}_gaussNewtonGPU_400")
    return _load(name).get(key)


def legal_runs(key, step=1, trig=True):
    """[(label, cost after `step` Gauss-Newton steps)] of every frozen run of workload `key` (e.g. "horizon_2048_float_400", "solve8_2048_float", "bench_4096_float_400x2").
    trig=False leaves out the runs with the seeded 1-ulp float sin / cos stand-in (`_trig`, `_trig_fma`): a SYNTHETIC perturbation, not an arithmetic the reference is known
    to take -- tests assert their bars with and without it."""
    runs = []
    for label, name in (("exact-order plain", "horizon_costs.json"), ("exact-order fma", "horizon_costs_fma.json")):
        e = _exact(key, name)
        if e and len(e["costs"]) > step:
            runs.append((label, e["costs"][step]))
    R = _reference_order()
    keys = [(key, "")]
    if key.startswith("bench_"):      # the first two steps of the frozen 8 x 400 solves are runs of the same workload
        _, size, prec, _ = key.split("_")
        keys.append((f"solve8_{size}_{prec}", " (8 x 400 solve)"))
    for k, note in keys:
        for sfx, tag in VARIANTS:
            if not trig and "trig" in tag:
                continue
            for seed, costs in sorted(R.get(k + sfx, {}).get("costs_by_seed", {}).items(), key=lambda kv: int(kv[0])):
                if len(costs) > step:
                    runs.append((f"reference-order {tag} seed {seed}{note}", costs[step]))
    return runs


def anchor(key, step=1):
    e = _exact(key, "horizon_costs.json")
    return e["costs"][step] if e and len(e["costs"]) > step else None


def spread(key, step=1, trig=True):
    """Diameter of the legal runs relative to the anchor; None without at least two runs."""
    runs = [c for _, c in legal_runs(key, step, trig)]
    a = anchor(key, step)
    if len(runs) < 2 or a is None:
        return None
    return (max(runs) - min(runs)) / abs(a)


def seed_spread(key, step=1):
    """Seed-to-seed diameter of the reference-order runs of the plain build alone (what two runs of the reference differ by, nothing else varied)."""
    R = _reference_order().get(key, {}).get("costs_by_seed", {})
    v = [c[step] for c in R.values() if len(c) > step]
    a = anchor(key, step)
    return (max(v) - min(v)) / abs(a) if len(v) >= 2 and a else None


def yardstick(key, precision, step=1, trig=True):
    s = spread(key, step, trig)
    return max(FLOOR[precision], s or 0.0)


def n_reference_order_runs(key):
    return sum(1 for label, _ in legal_runs(key) if label.startswith("reference-order"))


def exact_hull(key, step=1):
    """(low, high) of the two exact-order oracle runs of `key` (plain build, fused-multiply-add build): the same algorithm and sums under the two legal contractions of its
    elementwise arithmetic.  None if either is not frozen."""
    a, f = _exact(key, "horizon_costs.json"), _exact(key, "horizon_costs_fma.json")
    if not a or not f or len(a["costs"]) <= step or len(f["costs"]) <= step:
        return None
    return min(a["costs"][step], f["costs"][step]), max(a["costs"][step], f["costs"][step])


def progress_per_iteration(family, size, precision, L):
    """|c(L-1) - c(L+1)| / 2 of the exact-order plain oracle: what ONE more PCG iteration does to the cost at horizon L (frozen neighbours: make_horizon_costs.py --horizons)."""
    G = _load("horizon_costs.json")
    lo, hi = G.get(f"{family}_{size}_{precision}_{L - 1}"), G.get(f"{family}_{size}_{precision}_{L + 1}")
    if not lo or not hi:
        return None
    return abs(lo["costs"][1] - hi["costs"][1]) / 2.0


def iterations_from_hull(family, size, precision, L, cost):
    """How many PCG iterations of progress `cost` lies outside the hull of the two exact-order oracle runs at horizon L (0 inside); None if the neighbours are not frozen."""
    h, p = exact_hull(f"{family}_{size}_{precision}_{L}"), progress_per_iteration(family, size, precision, L)
    if h is None or not p:
        return None
    return max(0.0, h[0] - cost, cost - h[1]) / p


def verdict(key, precision, hip_cost, step=1):
    """{'distance', 'spread', 'yardstick', 'factor', 'within_reference_spread', 'within_contract', 'runs'} for one HIP cost."""
    a = anchor(key, step)
    if a is None:
        return None
    d = abs(hip_cost - a) / abs(a)
    y = yardstick(key, precision, step)
    return {"distance_from_exact_order_oracle": d, "reference_spread": spread(key, step), "seed_to_seed_spread": seed_spread(key, step), "yardstick": y, "factor": FACTOR,
            "within_reference_spread": d <= FACTOR * y, "within_contract": d <= FLOOR[precision], "legal_runs": len(legal_runs(key, step)),
            "reference_order_runs": n_reference_order_runs(key)}


def table(families=("horizon", "adversarial"), precisions=("float", "double"), horizons=(20, 50, 100, 200, 400)):
    rows = []
    for fam in families:
        size = 2048 if fam == "horizon" else 1024
        for prec in precisions:
            for L in horizons:
                key = f"{fam}_{size}_{prec}_{L}"
                if anchor(key) is None:
                    continue
                rows.append({"key": key, "family": fam, "precision": prec, "liters": L, "anchor": anchor(key), "runs": legal_runs(key), "spread": spread(key),
                             "seed_to_seed": seed_spread(key), "yardstick": yardstick(key, prec)})
    return rows


def _tail_note():
    key = "horizon_2048_float_50"; a = anchor(key)
    d = sorted(abs(c - a) / a for _, c in legal_runs(key))
    near = sum(1 for v in d if v <= 1e-5); far = [v for v in d if v > 1e-4]
    return (f"The outcomes are heavy-tailed: after 50 float iterations {near} of {len(d)} runs agree to 1e-5 and {len(far)} sit " + ", ".join(f"{v:.1e}" for v in far) +
            " away -- a handful of runs does not measure the spread.")


def markdown():
    f = lambda v: "n/a" if v is None else f"{v:.2e}"
    out = ["# Spread of the reference's own arithmetic (image_warping, one Gauss-Newton step, cost after L PCG iterations)", "",
           "Frozen oracle runs (tests/golden/reference_order_costs*.json, horizon_costs.json, horizon_costs_fma.json; generators beside them).  A *legal run* is the same",
           "algorithm under another rounding the reference itself may take: `reference-order` = its own sums (one float term per pixel, the 32-lane shfl.down tree of",
           "util.t:612-623, one float atomicAdd per warp -- solverGPUGaussNewton.t:312-317 -- committed in a seeded random order); `fma` = the restatement compiled with fused",
           "multiply-adds (what an NVPTX / AMDGPU back end does to the generated code); `raster` = J^T J p scattered in raster order instead of the banded order of the",
           "multi-threaded oracle; `exact-order` = sums accumulated in long double.  `seed-to-seed` = diameter of the reference-order runs of the plain build alone (nothing varied",
           "but the commit order of the atomics); `all legal runs` = diameter of everything; yardstick = max(contract floor 1e-5 / 1e-12, all legal runs).  " + _tail_note(), "",
           "| workload | precision | L | exact-order oracle cost | runs | seed-to-seed | all legal runs | yardstick |", "|---|---|---|---|---|---|---|---|"]
    for r in table():
        out.append(f"| {r['family']} | {r['precision']} | {r['liters']} | {r['anchor']:.9g} | {len(r['runs'])} | {f(r['seed_to_seed'])} | {f(r['spread'])} | {f(r['yardstick'])} |")
    out += ["", "## Multi-step solves: cost after every Gauss-Newton step (400 PCG iterations each), relative diameter of the legal runs", "",
            "| workload | precision | runs | " + " | ".join(f"step {i}" for i in range(1, 9)) + " |", "|---|---|---|" + "---|" * 8]
    for key, prec, steps in (("bench_4096_float_400x2", "float", 2), ("solve8_2048_float", "float", 8), ("solve8_2048_double", "double", 8), ("solve8_4096_float", "float", 8)):
        if anchor(key, 1) is None:
            continue
        cells = [f(spread(key, i)) if i <= steps else "" for i in range(1, 9)]
        out.append(f"| {key} | {prec} | {len(legal_runs(key, steps))} | " + " | ".join(cells) + " |")
    out += ["", "Individual runs at the short horizons (relative distance from the exact-order plain oracle):", ""]
    for L in (20, 50, 100):
        key = f"horizon_2048_float_{L}"; a = anchor(key)
        out.append(f"* L = {L}: " + ", ".join(f"{(c - a) / a:+.1e}" for _, c in legal_runs(key)))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    print(markdown())
