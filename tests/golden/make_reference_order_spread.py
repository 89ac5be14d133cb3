#!/usr/bin/env python
"""How reproducible is the REFERENCE against itself?  Frozen oracle runs in its "reference order" reduction mode (oracle/solver.hpp header:
per-element opt_float terms, the 32-lane shfl.down tree of API/src/util.t:612-623, one opt_float atomicAdd per warp --
API/src/solverGPUGaussNewton.t:312-317 -- in a seeded random order) for several seeds, on the workloads of tests/golden/make_horizon_costs.py:

  horizon      : image_warping 2048^2 float, ONE Gauss-Newton step with lIterations in {20, 50, 100, 200, 400};
  adversarial  : 1024^2 with 0.2 % stiff fit pixels (w_fit = 1e4, w_reg = 1e-4), same horizons;
  solve8       : the metric's solve at 2048^2 (--size 4096: at the metric's own size), 8 Gauss-Newton steps x 400 PCG iterations from the initial guess;
  bench        : what bench.py's timed region starts with, at --size (4096): 2 Gauss-Newton steps x 400 PCG iterations.

Every seed is one legal run of the reference's arithmetic (its atomics commit in an order the hardware does not define).  The seed-to-seed
spread of the cost at a horizon is what the reference's own trajectory contract can mean there; tests/test_horizon_gpu.py and bench.py read
the frozen spreads from tests/golden/reference_order_costs.json.  Oracle outputs, generated offline (minutes to hours of host time).

    python tests/golden/make_reference_order_spread.py [--seeds 1 2 3 4 5] [--families horizon adversarial solve8] [--threads 8] [--precisions float] [--variant fma] [--trig --tag trig]
"""
import argparse
import json
import os
import sys
import time

if "--variant" in sys.argv and sys.argv[sys.argv.index("--variant") + 1] == "fma":
    os.environ["OPT_ORACLE_VARIANT"] = "fma"      # read by oracle/binding.py at import: the same restatement compiled with fused multiply-adds allowed

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from opt_amd import workloads as wl          # noqa: E402
from oracle.binding import OracleSolver      # noqa: E402

OUT = os.path.join(HERE, "reference_order_costs.json")
HORIZONS = [20, 50, 100, 200, 400]
ADVERSARIAL = dict(fit_fraction=0.002, w_fit_sqrt=100.0, w_reg_sqrt=0.01, random_state=5)


def run(P, dbl, steps, liters, threads, seed):
    o = OracleSolver("image_warping", "gaussNewtonGPU", dbl, P.dims)
    o.set_threads(threads)
    o.set_reduction(1, seed)
    o.set("nIterations", steps); o.set("lIterations", liters)
    t0 = time.time()
    o.init(P.params)
    costs = [o.cost()]
    for _ in range(steps):
        if not o.step(P.params):
            break
        costs.append(o.cost())
    o.close()
    return costs, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2, 3, 4, 5])
    ap.add_argument("--families", nargs="+", default=["horizon", "adversarial", "solve8"])
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--precisions", nargs="+", default=["float"])
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--horizons", type=int, nargs="+", default=HORIZONS, help="horizon / adversarial families: which lIterations")
    ap.add_argument("--tag", default="", help="key suffix for another legal variant of the run, e.g. `raster` with --threads 1: the single-threaded oracle scatters J^T J p in raster "
                                               "order, the multi-threaded one in two colours of 4-row bands (oracle/solver.hpp forEachInstanceBanded) -- two accumulation orders of the same float additions")
    ap.add_argument("--out", default=OUT, help="JSON file to extend (tools/reference_spread.py reads reference_order_costs*.json)")
    ap.add_argument("--variant", default="plain", choices=["plain", "fma"], help="fma: keys get the suffix _fma")
    ap.add_argument("--trig", action="store_true", help="also vary the float sin / cos: seed n runs with oracle trig variant n (oracle/dual.hpp: a seeded implementation within 1 ulp of the "
                                                       "correctly rounded value -- the reference calls libdevice's __nv_sinf / __nv_cosf, util.t:162-171, this restatement the host libm); use with --tag trig")
    a = ap.parse_args()
    sfx = ("_" + a.tag if a.tag else "") + ("_fma" if a.variant == "fma" else "")
    out = a.out
    res = json.load(open(out)) if os.path.exists(out) else {}

    def done(key, seed):
        return str(seed) in res.get(key, {}).get("costs_by_seed", {})

    def put(key, seed, costs, dt):
        e = res.setdefault(key, {"costs_by_seed": {}, "seconds_by_seed": {}, "reduction": "reference order (oracle reductionMode 1)"})
        e["costs_by_seed"][str(seed)] = costs; e["seconds_by_seed"][str(seed)] = dt
        json.dump(res, open(out + ".tmp", "w"), indent=1, sort_keys=True)
        os.replace(out + ".tmp", out)
        print(key, "seed", seed, costs, f"{dt:.0f} s", flush=True)

    for fam in a.families:
        for prec in a.precisions:
            dbl = prec == "double"
            for seed in a.seeds:
                if a.trig:
                    from oracle import binding
                    binding.set_trig_variant(seed)
                if fam in ("horizon", "adversarial"):
                    for L in a.horizons:
                        size = a.size if fam == "horizon" else 1024
                        key = f"{fam}_{size}_{prec}_{L}{sfx}"
                        if done(key, seed):
                            continue
                        P = wl.image_warping(size, size, double=dbl) if fam == "horizon" else wl.image_warping(size, size, double=dbl, **ADVERSARIAL)
                        put(key, seed, *run(P, dbl, 1, L, a.threads, seed))
                elif fam == "bench":
                    key = f"bench_{a.size}_{prec}_400x2{sfx}"
                    if done(key, seed):
                        continue
                    put(key, seed, *run(wl.image_warping(a.size, a.size, double=dbl), dbl, 2, 400, a.threads, seed))
                else:
                    key = f"solve8_{a.size}_{prec}{sfx}"
                    if done(key, seed):
                        continue
                    put(key, seed, *run(wl.image_warping(a.size, a.size, double=dbl), dbl, 8, 400, a.threads, seed))


if __name__ == "__main__":
    main()
