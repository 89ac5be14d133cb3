"""Block-local patch solver vs the global PCG loop on poisson_image_editing: time and energy after equal numbers of sweeps /
PCG iterations.  usage: python tools/patch_bench.py [W]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from opt_amd import api, workloads as wl

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P = wl.poisson_image_editing(W, W, seed=0)
rows = []
for kind, extra in (("gaussNewtonGPU", {}), ("patchGaussNewtonGPU", {"patchSize": 32}), ("patchGaussNewtonGPU", {"patchSize": 16})):
    for warm in (True, False):
        dev = api.to_device(P)
        s = api.Solver(api.energy_file(P.energy), kind, P.dims, timing=not warm)
        s.set_parameter("nIterations", 4); s.set_parameter("lIterations", 16)
        for k, v in extra.items():
            s.set_parameter(k, v)
        torch.cuda.synchronize(); t = time.perf_counter()
        s.init(dev); c0 = s.cost()
        costs = []
        while s.step(dev):
            costs.append(s.cost())
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        if not warm:
            kt = s.kernel_timings()
            rows.append({"kind": kind, **extra, "wall_ms": dt * 1e3, "cost0": c0, "costs": costs,
                         "kernel_avg_us": {k: round(v[1] / v[0] * 1e3, 2) for k, v in kt.items()}})
        s.close()
for r in rows:
    print(json.dumps(r))
