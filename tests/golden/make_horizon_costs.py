#!/usr/bin/env python
"""Frozen ORACLE costs for the long-horizon parity experiments (tests/test_horizon_gpu.py, tools/horizon_parity.py, bench.py).

Three families, all image_warping / gaussNewtonGPU, float and double, written to tests/golden/horizon_costs.json:

  horizon      : the bench workload at 2048^2, ONE Gauss-Newton step with lIterations in {20, 50, 100, 200, 400}: the cost after the
                 step as a function of the PCG horizon.  The HIP loops (reference-ordered three-kernel loop, r-in-memory single kernel,
                 r-free single kernel) are compared with these at every horizon.
  adversarial  : 1024^2 with 0.2 % stiff fit pixels (w_fit = 1e4, w_reg = 1e-4): Jacobi preconditioner entries spanning eight
                 decades, where rebuilding r from stored search directions (r = (p_k - beta p_{k-1}) / M) is at its worst.  Same horizons.
  horizon --variant fma : the horizon family again with the oracle compiled with fused multiply-adds allowed (oracle/Makefile,
                 libopt_oracle_fma.so): a second legal rounding of the same algorithm, frozen in horizon_costs_fma.json.  |plain - fma| at a horizon is
                 what rounding alone does to the trajectory; it is the yardstick the HIP loops' distances are read against.
  solve8       : the metric's solve -- nIterations = 8, lIterations = 400 (examples/image_warping/src/main.cpp:113-114) from the initial
                 guess, cost after every step, at 2048^2 and 4096^2.

These are ORACLE outputs (CPU restatement of solverGPUGaussNewton.t), not reference outputs: the reference cannot run here
(DESIGN.md section 5).  Row-banded OpenMP mode with a fixed thread count (recorded) so the sums are reproducible.

    python tests/golden/make_horizon_costs.py [--families horizon adversarial solve8] [--sizes 2048 4096] [--threads 8]
"""
import argparse
import json
import os
import sys
import time

if "--variant" in sys.argv and sys.argv[sys.argv.index("--variant") + 1] == "fma":
    os.environ["OPT_ORACLE_VARIANT"] = "fma"      # read by oracle/binding.py at import

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from opt_amd import workloads as wl          # noqa: E402
from oracle.binding import OracleSolver      # noqa: E402

OUT = os.path.join(HERE, "horizon_costs_fma.json" if os.environ.get("OPT_ORACLE_VARIANT") == "fma" else "horizon_costs.json")
HORIZONS = [20, 50, 100, 200, 400]
ADVERSARIAL = dict(fit_fraction=0.002, w_fit_sqrt=100.0, w_reg_sqrt=0.01, random_state=5)
ADVERSARIAL_SIZE = 1024


def run(P, dbl, steps, liters, threads):
    o = OracleSolver("image_warping", "gaussNewtonGPU", dbl, P.dims)
    o.set_threads(threads)
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


def save(res):
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", nargs="+", default=["horizon", "adversarial", "solve8"])
    ap.add_argument("--sizes", type=int, nargs="+", default=[2048, 4096], help="solve8 sizes")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--precisions", nargs="+", default=["float", "double"])
    ap.add_argument("--redo", action="store_true")
    ap.add_argument("--horizons", type=int, nargs="+", default=HORIZONS, help="horizon / adversarial families: which lIterations (round 5: 26, 37, 48 = quiet\n"
                    "iterations of the horizon family's ~5.5-iteration amplification cycle, profiles/r05_l50_bisect.md)")
    ap.add_argument("--variant", default="plain", choices=["plain", "fma"])
    args = ap.parse_args()
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for fam in args.families:
        for prec in args.precisions:
            dbl = prec == "double"
            if fam in ("horizon", "adversarial"):
                for L in args.horizons:
                    key = f"{fam}_{2048 if fam == 'horizon' else ADVERSARIAL_SIZE}_{prec}_{L}"
                    if key in res and not args.redo:
                        continue
                    P = wl.image_warping(2048, 2048, double=dbl) if fam == "horizon" else \
                        wl.image_warping(ADVERSARIAL_SIZE, ADVERSARIAL_SIZE, double=dbl, **ADVERSARIAL)
                    costs, dt = run(P, dbl, 1, L, args.threads)
                    res[key] = {"costs": costs, "threads": args.threads, "seconds": dt}
                    print(key, costs, f"{dt:.0f} s", flush=True)
                    save(res)
            elif fam == "solve8":
                for size in args.sizes:
                    key = f"solve8_{size}_{prec}"
                    if key in res and not args.redo:
                        continue
                    costs, dt = run(wl.image_warping(size, size, double=dbl), dbl, 8, 400, args.threads)
                    res[key] = {"costs": costs, "threads": args.threads, "seconds": dt}
                    print(key, costs, f"{dt:.0f} s", flush=True)
                    save(res)


if __name__ == "__main__":
    main()
