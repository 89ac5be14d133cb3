#!/usr/bin/env python
"""Rows marched per workgroup on a slab-sized problem: fewer, taller row groups re-read fewer halo rows (4 per group) but leave CUs idle.
    python tools/slab_rows.py     -> us per PCG iteration of plain 4096x512 / 4096x1024 problems for OPT_AMD_ITER_ROWS in {natural, ...}"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
from opt_amd import api, workloads as wl       # noqa: E402


def run(W, H, rows, liters=400, steps=3):
    if rows: os.environ["OPT_AMD_ITER_ROWS"] = str(rows)
    else: os.environ.pop("OPT_AMD_ITER_ROWS", None)
    P = wl.image_warping(W, H)
    dev = api.to_device(P)
    s = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", (W, H))
    s.set_parameter("nIterations", steps + 1); s.set_parameter("lIterations", liters)
    s.init(dev); s.step(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step(dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = s.cost()
    s.close()
    return dt / (steps * liters) * 1e6, c


for (W, H) in [(4096, 512 + 16), (4096, 1024 + 16)]:
    for rep in range(2):
        for rows in (0, 14, 16, 19, 22, 26, 32, 44):
            us, c = run(W, H, rows)
            print(f"{W}x{H} rows/group {rows or 'natural':>7}: {us:6.1f} us/iter  cost {c:.6g}", flush=True)
