"""Development aid: one ARAP Gauss-Newton solve on the GPU next to the oracle, printed step by step (python -u tools/round5/dbg_arap.py nx ny [liters] [steps])."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from opt_amd import api, workloads as wl
from oracle import binding
from helpers import device_unknowns, flat_unknowns, hip_solver, oracle_solver, rel_err

nx, ny = int(sys.argv[1]), int(sys.argv[2])
liters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
P = wl.arap_mesh_deformation(nx, ny, seed=5, perturb=0.01)
o = oracle_solver(binding, P, "gaussNewtonGPU", nIterations=steps, lIterations=liters)
g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=steps, lIterations=liters)
dev = api.to_device(P)
Pref = P.clone()
o.init(Pref.params); g.init(dev)
print("init", o.cost(), g.cost(), flush=True)
while True:
    a, b = o.step(Pref.params), g.step(dev)
    print("step", a, b, o.cost(), g.cost(), abs(g.cost() - o.cost()) / abs(o.cost()), "status", g.on_chip_status(), flush=True)
    if not a or not b:
        break
print(sorted(g.kernel_timings().keys()), flush=True)
print("x err", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), flush=True)
