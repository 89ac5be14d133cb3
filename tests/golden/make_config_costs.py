#!/usr/bin/env python
"""Frozen ORACLE cost histories of BASELINE configs 3 and 4 at their OWN horizons:

  config 3  shape_from_shading 1024 x 1024 double, LMGPU, nIterations 60 x lIterations 10   (shape_from_shading/src/main.cpp:27-38)
  config 4  arap_mesh_deformation 708 x 707 grid mesh (500 556 vertices) float, gaussNewtonGPU, 20 x 100   (arap_mesh_deformation/src/main.cpp:75-79)

The inputs are the synthetic workloads bench / tests use (opt_amd/workloads.py), so nothing but the seeds travels.  Per outer step the file keeps
the accepted cost (Opt_ProblemCurrentCost), for LM the trust-region radius, and for config 4 also a DOUBLE run of the same inputs (the rounding-free yardstick of the
float trajectory).  tests/test_config_horizon_gpu.py steps both HIP paths (default loop, reference-ordered loop) side by side with these numbers and reports where and
at what rate they part.  These are ORACLE outputs, not reference outputs (the reference is Terra -> PTX and cannot run here).

    python tests/golden/make_config_costs.py [--configs 3 4] [--threads 8]
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

OUT = os.path.join(HERE, "config_costs.json")


def run(P, kind, n_it, l_it, threads, lm):
    o = OracleSolver(P.energy, kind, P.double, P.dims)
    o.set_threads(threads)
    o.set("nIterations", n_it); o.set("lIterations", l_it)
    t0 = time.time()
    o.init(P.params)
    costs, radii = [o.cost()], [o.trust_region_radius() if lm else 0.0]
    while o.step(P.params):
        costs.append(o.cost()); radii.append(o.trust_region_radius() if lm else 0.0)
        print(P.energy, kind, len(costs) - 1, repr(costs[-1]), f"{time.time() - t0:.0f} s", flush=True)
    # (a step that returns 0 after the function-tolerance exit leaves prevCost as it was: nothing to append)
    out = {"costs": costs, "steps_taken": len(costs) - 1, "threads": threads, "seconds": time.time() - t0}
    if lm:
        out["radii"] = radii
    o.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, nargs="+", default=[3, 4])
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    if 3 in args.configs:
        P = wl.shape_from_shading(1024, 1024, double=True, holes=True)
        res["config3_sfs_1024_double_LM_60x10"] = dict(run(P, "LMGPU", 60, 10, args.threads, True), workload="workloads.shape_from_shading(1024, 1024, double=True, holes=True)")
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    if 4 in args.configs:
        for dbl in (False, True):
            P = wl.arap_mesh_deformation(708, 707, perturb=0.01, double=dbl)
            key = f"config4_arap_708x707_{'double' if dbl else 'float'}_GN_20x100"
            res[key] = dict(run(P, "gaussNewtonGPU", 20, 100, args.threads, False), workload=f"workloads.arap_mesh_deformation(708, 707, perturb=0.01, double={dbl})")
            json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
