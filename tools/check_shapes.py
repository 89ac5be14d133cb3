#!/usr/bin/env python
"""Development check: PCG iteration rate of image_warping (GN, float) for different image SHAPES of the same pixel count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opt_amd import api, workloads as wl
for W, H in [(2048, 4096), (4096, 2048), (8192, 1024), (16384, 512), (8192, 8192)]:
    P = wl.image_warping(W, H)
    dev = api.to_device(P)
    s = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", (W, H))
    s.set_parameter("nIterations", 3); s.set_parameter("lIterations", 200)
    s.init(dev); s.step(dev); torch.cuda.synchronize(); t0 = time.perf_counter()
    s.step(dev); s.step(dev); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%5d x %5d: %.0f PCG it/s, %.1f Gpixel-iterations/s" % (W, H, 400 / dt, 400 / dt * W * H / 1e9), flush=True)
    s.close(); del dev
