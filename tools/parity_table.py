#!/usr/bin/env python
"""Cost parity of the image_warping Gauss-Newton loop variants against the float CPU oracle at short horizons (where the 1e-5 contract is meaningful):
r-free ring (default), r and p in memory with the deferred delta term rebuilt (OPT_AMD_RFREE=0), and the plain double-buffered loop (+ OPT_AMD_RECON_P=0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opt_amd import api, workloads as wl
from oracle.binding import OracleSolver
cases = [("96x64 2 GN x 8 PCG", lambda: wl.image_warping(96, 64, random_state=1, mask_fraction=0.05, perturb=0.3), 2, 8),
         ("300x200 2 x 20", lambda: wl.image_warping(300, 200, random_state=2, mask_fraction=0.05, perturb=0.3), 2, 20),
         ("1024x768 1 x 20", lambda: wl.image_warping(1024, 768, random_state=3, mask_fraction=0.05, perturb=0.3), 1, 20)]
print("| case | r-free (default) | RFREE=0 (p_{k-2} rebuilt) | RFREE=0 RECON_P=0 |\n|---|---|---|---|")
for name, make, n, l in cases:
    P = make(); ref = P.clone()
    o = OracleSolver(P.energy, "gaussNewtonGPU", False, P.dims); o.set_threads(64)
    o.set("nIterations", n); o.set("lIterations", l); o.solve(ref.params)
    row = f"| {name} |"
    for rf, rp in ((1, 1), (0, 1), (0, 0)):
        os.environ["OPT_AMD_RFREE"] = str(rf); os.environ["OPT_AMD_RECON_P"] = str(rp)
        dev = api.to_device(P)
        g = api.Solver(api.energy_file(P.energy), "gaussNewtonGPU", P.dims)
        g.set_parameter("nIterations", n); g.set_parameter("lIterations", l); g.solve(dev)
        row += f" {abs(g.cost() - o.cost()) / o.cost():.2e} |"
        g.close()
    print(row, flush=True)
