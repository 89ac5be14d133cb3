#!/usr/bin/env python
"""Golden scalars for bench.py's workload: the CPU oracle's cost trajectory on image_warping 4096^2 (and 2048^2, BASELINE
config 2) with the reference example's iteration counts (400 PCG iterations per Gauss-Newton step, main.cpp:113-114).

The oracle needs ~0.5 s per PCG iteration at 4096^2 on 128 threads, so the full-length trajectory cannot be recomputed
inside a GPU test; it is generated here once (float, as the metric is quoted, and double as the rounding-free yardstick) and
frozen in tests/golden/bench_costs.json.  bench.py prints the HIP path's cost after its first step next to these numbers and
tests/test_steady_state_gpu.py asserts the 1e-5 contract against them.  These are ORACLE outputs, not reference outputs
(the reference cannot run here).

    python tests/golden/make_bench_cost.py [--sizes 2048 4096] [--steps 2] [--threads 8]
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from opt_amd import workloads as wl          # noqa: E402
from oracle.binding import OracleSolver      # noqa: E402

OUT = os.path.join(HERE, "bench_costs.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[2048, 4096])
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--liters", type=int, default=400)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--kinds", nargs="+", default=["gaussNewtonGPU"])
    ap.add_argument("--precisions", nargs="+", default=["float", "double"])
    args = ap.parse_args()
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for size in args.sizes:
        for kind in args.kinds:
            for prec in args.precisions:
                dbl = prec == "double"
                key = f"image_warping_{size}x{size}_{prec}_{kind}_{args.liters}"
                P = wl.image_warping(size, size, double=dbl)
                o = OracleSolver("image_warping", kind, dbl, P.dims)
                o.set_threads(args.threads)
                o.set("nIterations", args.steps); o.set("lIterations", args.liters)
                t0 = time.time()
                o.init(P.params)
                costs = [o.cost()]
                for _ in range(args.steps):
                    if not o.step(P.params):
                        break
                    costs.append(o.cost())
                    print(key, costs, f"{time.time() - t0:.0f} s", flush=True)
                res[key] = {"costs": costs, "threads": args.threads, "seconds": time.time() - t0}
                o.close()
                json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
