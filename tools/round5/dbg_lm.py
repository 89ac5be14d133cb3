"""Development aid: one LM / GN solve of image_warping on the GPU next to the oracle, printed step by step (python -u tools/round5/dbg_lm.py [kind] [W] [H] [double] [liters] [period])."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from opt_amd import api, workloads as wl
from oracle import binding
from helpers import device_unknowns, flat_unknowns, hip_solver, oracle_solver, rel_err

kind = sys.argv[1] if len(sys.argv) > 1 else "LMGPU"
W, H = int(sys.argv[2]) if len(sys.argv) > 2 else 96, int(sys.argv[3]) if len(sys.argv) > 3 else 64
dbl = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
liters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
period = int(sys.argv[6]) if len(sys.argv) > 6 else 10
seed = int(sys.argv[7]) if len(sys.argv) > 7 else 5
maskf = float(sys.argv[8]) if len(sys.argv) > 8 else 0.1
print("building problem", kind, W, H, dbl, liters, period, flush=True)
P = wl.image_warping(W, H, double=dbl, random_state=seed, mask_fraction=maskf, perturb=0.4)
kw = dict(nIterations=3, lIterations=liters)
if kind == "LMGPU":
    kw["residual_reset_period"] = period
o = oracle_solver(binding, P, kind, **kw)
g = hip_solver(P, kind, timing=True, **kw)
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
