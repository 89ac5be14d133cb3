"""One Gauss-Newton step of 50 PCG iterations of image_warping 4096^2 with amd_reference_order = 1 (the profiled command of tools/round6/contract_pmc.sh)."""
import sys
sys.path.insert(0, ".")
import torch
from opt_amd import api, workloads as wl
P = wl.image_warping(4096, 4096)
dev = api.to_device(P)
s = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", P.dims)
s.set_parameter("amd_reference_order", 1); s.set_parameter("nIterations", 1); s.set_parameter("lIterations", 50)
s.init(dev); s.step(dev); torch.cuda.synchronize()
print("cost", s.cost()); s.close()
