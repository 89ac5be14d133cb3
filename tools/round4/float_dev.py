import sys, os, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import importlib
from opt_amd import api
import oracle.binding as ob
from helpers import hip_solver
te = importlib.import_module("test_energies_gpu")
for name in ("curveFitting", "cotangent", "intrinsic"):
    for kind in ("gaussNewtonGPU", "LMGPU"):
        P = te.CASES[name](False)
        o = ob.OracleSolver(P.energy, kind, P.double, P.dims); o.set("nIterations", 4); o.set("lIterations", 12)
        g = hip_solver(P, kind, nIterations=4, lIterations=12)
        Q = P.clone(); dev = api.to_device(P)
        o.init(Q.params); g.init(dev); s = max(abs(o.cost()), 1e-300); d = []
        while True:
            a, b = o.step(Q.params), g.step(dev)
            d.append(abs(g.cost() - o.cost()) / max(abs(o.cost()), 1e-7 * s))
            if not a or not b: break
        print(name, kind, ['%.1e' % v for v in d], flush=True)
        g.close(); o.close()
