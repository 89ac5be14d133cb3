#!/usr/bin/env python
"""Development check: image_warping on a slab-shaped wide image (8192 x 516, what one of 8 ranks holds at 8192^2 with 2 ghost rows):
the single-kernel iteration against the three-kernel loop, and the iteration rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opt_amd import api, workloads as wl

def solve(env, W, H, lit):
    for k in ("OPT_AMD_ONEKERNEL",): os.environ.pop(k, None)
    os.environ.update(env)
    P = wl.image_warping(W, H, random_state=5, perturb=0.3)
    dev = api.to_device(P)
    s = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", (W, H))
    s.set_parameter("nIterations", 2); s.set_parameter("lIterations", lit)
    s.init(dev); torch.cuda.synchronize(); t0 = time.perf_counter()
    while s.step(dev): pass
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    x = torch.cat([dev[0].reshape(-1), dev[1].reshape(-1)]).cpu().numpy().astype(np.float64); c = s.cost(); s.close()
    return x, c, dt

W, H, lit = 8192, 516, 40
x1, c1, t1 = solve({}, W, H, lit)
x3, c3, t3 = solve({"OPT_AMD_ONEKERNEL": "0"}, W, H, lit)
print("cost", c1, c3, "rel", abs(c1 - c3) / abs(c3), "x relerr", np.linalg.norm(x1 - x3) / np.linalg.norm(x3), "it/s one-kernel %.0f three-kernel %.0f" % (2 * lit / t1, 2 * lit / t3))
assert abs(c1 - c3) <= 2e-5 * abs(c3)
