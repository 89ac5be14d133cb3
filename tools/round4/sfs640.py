import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from opt_amd import api, workloads as wl
for rep in range(3):
    P = wl.shape_from_shading(640, 480, double=True)
    g = api.Solver(api.energy_file(P.energy), "LMGPU", P.dims, double=True)
    g.set_parameter("nIterations", 60); g.set_parameter("lIterations", 10)
    dev = api.to_device(P)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.solve(dev)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(os.environ.get("OPT_AMD_LIB", "default")[-14:], "640x480 LM 60x10: %.2f ms" % (dt * 1e3), g.cost(), flush=True)
    g.close()
