#!/usr/bin/env python
"""Kernel variants on a slab-sized problem (the 1/8 slab of 4096^2 is VALU / latency bound per workgroup, not HBM bound): us per PCG iteration of plain
4096x528 / 4096x1040 problems under the A/B switches of the image_warping energy."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
from opt_amd import api, workloads as wl       # noqa: E402

KEYS = ["OPT_AMD_RECOMPUTE_AP", "OPT_AMD_RFREE", "OPT_AMD_PAIR_DELTA", "OPT_AMD_SWEEP", "OPT_AMD_ITER_ROWS"]


def run(W, H, env, liters=400, steps=3):
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update(env)
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


VARIANTS = [{}, {"OPT_AMD_RECOMPUTE_AP": "0"}, {"OPT_AMD_RFREE": "0"}, {"OPT_AMD_RFREE": "0", "OPT_AMD_PAIR_DELTA": "0"}, {"OPT_AMD_SWEEP": "0"}]
for (W, H) in [(4096, 528), (4096, 1040), (4096, 4096)]:
    for rep in range(2):
        for env in VARIANTS:
            us, c = run(W, H, env, steps=3 if H < 4096 else 1)
            print(f"{W}x{H} {str(env):60s}: {us:6.1f} us/iter  cost {c:.6g}", flush=True)
