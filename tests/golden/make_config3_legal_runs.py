#!/usr/bin/env python
"""Legal oracle runs of BASELINE config 3 (shape_from_shading 1024^2 double LM 60 x 10) beside the exact-order plain run of make_config_costs.py: the same algorithm under
other roundings the reference itself may take -- the fused-multiply-add build of the oracle (run with OPT_ORACLE_VARIANT=fma) and the reference's own sums (oracle
reductionMode 1: per-warp shfl tree + one atomicAdd per warp committed in a seeded order, API/src/util.t:612-623) under two seeds.  Their per-step distance from the plain
run is the yardstick for how far ANY faithful implementation can be expected to follow this 60-step Levenberg-Marquardt trajectory (tests/test_config_horizon_gpu.py).

    python tests/golden/make_config3_legal_runs.py --seeds 1 2                    (plain build, reference-order sums)
    OPT_ORACLE_VARIANT=fma python tests/golden/make_config3_legal_runs.py --seeds 0        (fma build, exact-order sums; seed 0 = exact order)
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

OUT = os.path.join(HERE, "config3_legal_runs.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 2])
    ap.add_argument("--threads", type=int, default=2)
    args = ap.parse_args()
    build = "fma" if os.environ.get("OPT_ORACLE_VARIANT") == "fma" else "plain"
    P = wl.shape_from_shading(1024, 1024, double=True, holes=True)
    for seed in args.seeds:
        o = OracleSolver(P.energy, "LMGPU", True, P.dims)
        o.set_threads(args.threads)
        if seed:
            o.set_reduction(1, seed)
        o.set("nIterations", 60); o.set("lIterations", 10)
        Q = P.clone()
        t0 = time.time()
        o.init(Q.params)
        costs, radii = [o.cost()], [o.trust_region_radius()]
        while o.step(Q.params):
            costs.append(o.cost()); radii.append(o.trust_region_radius())
        o.close()
        res = json.load(open(OUT)) if os.path.exists(OUT) else {}
        key = f"config3_sfs_1024_double_LM_60x10|{build}|{'exact-order' if seed == 0 else 'reference-order seed %d' % seed}"
        res[key] = {"costs": costs, "radii": radii, "steps_taken": len(costs) - 1, "seconds": time.time() - t0}
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
        print(key, len(costs) - 1, costs[-1], f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
