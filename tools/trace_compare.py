#!/usr/bin/env python
"""Per-iteration PCG scalars (alphaNumerator, alphaDenominator, betaNumerator) of the HIP loops next to the oracle's on one problem: where do two loops part?

    python tools/trace_compare.py [--family adversarial|horizon] [--liters 30] [--double]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def hip_trace(family, dbl, liters, env):
    import horizon_parity as hp
    from opt_amd import api
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        P = hp.problem(family, dbl)
        dev = api.to_device(P)
        s = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", P.dims, double=dbl)
        s.set_parameter("nIterations", 1); s.set_parameter("lIterations", liters)
        s.enable_trace(True)
        s.init(dev); s.step(dev)
        t = s.trace(); c = s.cost()
        s.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return t, c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="adversarial")
    ap.add_argument("--liters", type=int, default=30)
    ap.add_argument("--double", action="store_true")
    args = ap.parse_args()
    import horizon_parity as hp
    from oracle.binding import OracleSolver
    P = hp.problem(args.family, args.double)
    o = OracleSolver("image_warping", "gaussNewtonGPU", args.double, P.dims)
    o.set_threads(min(os.cpu_count() or 1, 64))
    o.set("nIterations", 1); o.set("lIterations", args.liters)
    o.init(P.params); o.step(P.params)
    to, co = o.trace(), o.cost()
    traces = {"oracle": (to, co)}
    for name, env in hp.LOOPS.items():
        traces[name] = hip_trace(args.family, args.double, args.liters, env)
    print("cost after the step: " + "  ".join(f"{k} {v[1]:.9g}" for k, v in traces.items()))
    print("lIter | " + " | ".join(f"{k:>36s}" for k in traces))
    print("      | " + " | ".join(f"{'aNum':>11s} {'aDen':>11s} {'bNum':>11s}" for _ in traces))
    n = min(len(v[0]) for v in traces.values())
    for i in range(n):
        print(f"{i:5d} | " + " | ".join(f"{v[0][i][2]:11.5e} {v[0][i][3]:11.5e} {v[0][i][4]:11.5e}" for v in traces.values()))
    print("relative difference of betaNumerator / alphaNumerator (= beta) from the oracle's:")
    for i in range(n):
        bo = to[i][4] / to[i][2] if to[i][2] else 0.0
        print(f"{i:5d} | beta oracle {bo:11.5e} | " + " | ".join(
            f"{k} {abs((v[0][i][4] / v[0][i][2] if v[0][i][2] else 0.0) - bo) / abs(bo) if bo else 0.0:9.2e}" for k, v in traces.items() if k != "oracle"))


if __name__ == "__main__":
    main()
