import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from opt_amd import api, workloads as wl
SZ = int(sys.argv[2]) if len(sys.argv) > 2 else 512
P = wl.image_warping(SZ, SZ)
g = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", P.dims, timing=True)
g.set_parameter("nIterations", 14); g.set_parameter("lIterations", 10)
dev = api.to_device(P)
g.init(dev); g.step(dev)
torch.cuda.synchronize()
print("status after step 1", g.on_chip_status(), g.describe().get("path"), g.describe().get("workgroups"))
side = torch.cuda.Stream()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
print("occupy", api.lib().OptAmd_DebugOccupy(n, ctypes.c_double(300.0), ctypes.c_void_p(side.cuda_stream)))
time.sleep(0.02)
g.set_timing(1)
t0 = time.perf_counter(); g.step(dev); dt = time.perf_counter() - t0
print("step 2", dt, "status", g.on_chip_status())
for k, v in g.kernel_timings().items():
    print("   %-24s n=%3d total %9.3f ms" % (k, v[0], v[1]))
side.synchronize()
g.set_timing(1)
t0 = time.perf_counter(); g.step(dev); dt = time.perf_counter() - t0
print("step 3 (tenant gone)", dt, "status", g.on_chip_status())
for k, v in g.kernel_timings().items():
    print("   %-24s n=%3d total %9.3f ms" % (k, v[0], v[1]))
