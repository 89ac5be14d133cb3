#!/usr/bin/env python
"""Frozen oracle runs of the reference's FULL default image_warping flow (examples/image_warping/src/main.cpp:98-139, CombinedSolver.h:150-207):
512^2 unit-lattice image, border pinned, the nine cat512 markers ramped from source to target over 19 passes, each pass one solve of
8 non-linear x 400 PCG iterations -- exactly what examples/image_warping_example.cpp drives through the C API (same float arithmetic for the
constraint image).  For Gauss-Newton and Levenberg-Marquardt, float, in

  * the exact-order mode of the oracle (long-double sums, rounded once), and
  * its reference-order mode (oracle/solver.hpp reductionMode 1: one float term per pixel, the 32-lane shfl.down tree, one float atomicAdd per warp in a
    seeded random order) for several seeds: every seed is one legal run of the reference's own arithmetic.

tests/test_cpp_callers_gpu.py::test_image_warping_reference_flow_against_frozen_runs and bench.py's `reference_example_flows` compare the HIP paths' final
costs with these.  Oracle outputs, generated offline (about an hour of host time per run on 8 cores); the file is extended run by run, so the script
can be interrupted and restarted.

    python tests/golden/make_reference_flow.py [--kinds gaussNewtonGPU LMGPU] [--seeds 0 1 2 3 4 5] [--threads 8] [--size 512] [--passes 19] [--nl 8] [--li 400]
    (seed 0 = the exact-order run)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from opt_amd import workloads as wl          # noqa: E402
from oracle.binding import OracleSolver      # noqa: E402

OUT = os.path.join(HERE, "reference_flow_costs.json")
MARKERS = [(30, 132, 59, 44), (229, 51, 157, 91), (430, 124, 379, 42), (281, 369, 326, 323), (197, 407, 163, 418),
           (64, 386, 26, 300), (311, 168, 253, 182), (89, 228, 56, 255), (92, 192, 84, 192)]      # cat512.constraints (image_warping/src/main.cpp:4-27 reads them)


def ramp_flow(size, passes, nl, li, kind, dbl, threads, seed, progress=None):
    """The constraint ramp of examples/image_warping_example.cpp on the oracle; returns the cost after every pass."""
    ft = np.float64 if dbl else np.float32
    P = wl.image_warping(size, size, double=dbl)
    P.params[3][...] = -1
    xs = np.arange(size, dtype=ft)
    P.params[3][0, :, 0] = xs; P.params[3][0, :, 1] = 0
    P.params[3][size - 1, :, 0] = xs; P.params[3][size - 1, :, 1] = size - 1
    P.params[3][:, 0, 0] = 0; P.params[3][:, 0, 1] = xs
    P.params[3][:, size - 1, 0] = size - 1; P.params[3][:, size - 1, 1] = xs
    P.params[0][...] = P.params[2]; P.params[1][...] = 0; P.params[4][...] = 0
    o = OracleSolver("image_warping", kind, dbl, P.dims)
    o.set_threads(threads)
    if seed:
        o.set_reduction(1, seed)
    o.set("nIterations", nl); o.set("lIterations", li)
    costs = []
    for i in range(passes):
        alpha = np.float32(i + 1) / np.float32(passes)
        for (x0, y0, x1, y1) in MARKERS:
            x, y = x0 * size // 512, y0 * size // 512
            tx, ty = np.float32(x1 * size // 512), np.float32(y1 * size // 512)
            P.params[3][y, x] = (ft((np.float32(1) - alpha) * np.float32(x) + alpha * tx), ft((np.float32(1) - alpha) * np.float32(y) + alpha * ty))
        o.solve(P.params)
        costs.append(o.cost())
        if progress:
            progress(i, costs[-1])
    o.close()
    return costs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", nargs="+", default=["gaussNewtonGPU", "LMGPU"])
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2, 3, 4, 5])
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--passes", type=int, default=19)
    ap.add_argument("--nl", type=int, default=8)
    ap.add_argument("--li", type=int, default=400)
    ap.add_argument("--out", default=OUT)
    a = ap.parse_args()
    res = json.load(open(a.out)) if os.path.exists(a.out) else {}
    for seed in a.seeds:
        for kind in a.kinds:
            key = f"image_warping_{a.size}_float_{kind}_{a.passes}x{a.nl}x{a.li}"
            e = res.setdefault(key, {"costs_by_seed": {}, "seconds_by_seed": {}, "note": "seed 0: exact-order sums; seed n > 0: reference-order sums (oracle reductionMode 1), cost after every pass"})
            if str(seed) in e["costs_by_seed"]:
                continue
            t0 = time.time()
            costs = ramp_flow(a.size, a.passes, a.nl, a.li, kind, False, a.threads, seed,
                              progress=lambda i, c: print(f"  {key} seed {seed} pass {i}: {c:.6f} ({time.time() - t0:.0f} s)", flush=True))
            e["costs_by_seed"][str(seed)] = costs; e["seconds_by_seed"][str(seed)] = time.time() - t0
            json.dump(res, open(a.out + ".tmp", "w"), indent=1, sort_keys=True)
            os.replace(a.out + ".tmp", a.out)
            print(key, "seed", seed, "final", costs[-1], f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
