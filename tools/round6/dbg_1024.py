import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from opt_amd import api, workloads as wl
from oracle.binding import OracleSolver
P = wl.image_warping(1024, 1024)
o = OracleSolver("image_warping", "gaussNewtonGPU", False, P.dims); o.set_threads(16)
o.set("nIterations", 3); o.set("lIterations", 10)
Pr = P.clone(); o.init(Pr.params); oc = [o.cost()]
while o.step(Pr.params): oc.append(o.cost())
print("oracle   ", oc)
for name, kw in (("onchip", {}), ("streaming", {"amd_onchip": 0}), ("ref-order", {"amd_reference_order": 1})):
    g = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", P.dims)
    g.set_parameter("nIterations", 3); g.set_parameter("lIterations", 10)
    for k, v in kw.items(): g.set_parameter(k, v)
    dev = api.to_device(P); g.init(dev); c = [g.cost()]
    while g.step(dev): c.append(g.cost())
    print("%-10s" % name, c, [abs(a - b) / abs(b) for a, b in zip(c, oc)], g.on_chip_status()); g.close()
# mixed: step 1 on chip, step 2+ streaming (what a fall-back does)
g = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", P.dims)
g.set_parameter("nIterations", 3); g.set_parameter("lIterations", 10)
dev = api.to_device(P); g.init(dev); c = [g.cost()]
g.step(dev); c.append(g.cost()); g.set_parameter("amd_onchip", 0)
while g.step(dev): c.append(g.cost())
print("mixed     ", c, [abs(a - b) / abs(b) for a, b in zip(c, oc)]); g.close()
