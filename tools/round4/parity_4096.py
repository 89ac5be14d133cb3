"""Cost after one and two Gauss-Newton steps of 400 PCG iterations at 4096^2 (bench.py's first steps) for the HIP loops, against the frozen exact-order oracle
(tests/golden/bench_costs.json): is the benchmarked loop's distance a property of the reformulation, or of float rounding in any order?"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from opt_amd import api, workloads as wl

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "bench_costs.json")))
ref = G["image_warping_4096x4096_float_gaussNewtonGPU_400"]["costs"]
dbl = G["image_warping_4096x4096_double_gaussNewtonGPU_400"]["costs"]
for name, env in (("r-free (benchmarked)", {}), ("reference-ordered three-kernel loop", {"OPT_AMD_ONEKERNEL": "0"})):
    for k, v in env.items(): os.environ[k] = v
    P = wl.image_warping(4096, 4096)
    dev = api.to_device(P)
    s = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", P.dims)
    s.set_parameter("nIterations", 2); s.set_parameter("lIterations", 400)
    s.init(dev); c = [s.cost()]
    while s.step(dev): c.append(s.cost())
    c.append(s.cost()); s.close()
    for k in env: os.environ.pop(k)
    print(name, c[:3], "rel vs float oracle", [abs(a - b) / abs(b) for a, b in zip(c[:3], ref)], flush=True)
print("float oracle", ref, "double oracle", dbl, "float vs double", [abs(a - b) / abs(b) for a, b in zip(ref, dbl)])
