#!/usr/bin/env python
"""How far apart do LEGAL float runs of the reference's arithmetic end on the small trajectory cases of tests/test_energies_gpu.py?

The float trajectories of some example energies (intrinsic_image_decomposition: ill-conditioned, L_p re-weighting; the graph energies: float atomics scattered in an
undefined order; curveFitting: cos / sin of ~600 rad) do not hold the 1e-5 contract between ANY two implementations.  Instead of a hand-set bar, test_trajectory takes the
diameter of a set of frozen legal oracle runs of the same case (never below the contract), like tests/test_horizon_gpu.py does for image_warping:

    exact plain / exact fma        sums in long double; the restatement compiled without / with fused multiply-adds (oracle/Makefile)
    reference-order seeds          oracle reductionMode 1: image energies -- the reference's warp tree + per-warp float atomics in a seeded order; graph energies -- the
                                   hyperedges scattered in a seeded random order (oracle/solver.hpp header)

Output: tests/golden/float_envelopes.json  {case_kind: {"runs": {label: [cost after init, step 1, ...]}, "x_spread": largest relative distance of the final unknowns}}.
Runs both builds of the oracle in subprocesses (the binding loads one library per process).  Seconds of host time.

    python tests/golden/make_float_envelopes.py
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "float_envelopes.json")
NAMES = ["curveFitting", "cotangent", "embedded", "embedded_rest", "robust", "intrinsic", "arap", "flow", "poisson"]
KINDS = ["gaussNewtonGPU", "LMGPU"]
SEEDS = [1, 2, 3, 4]


def worker():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import numpy as np
    import oracle.binding as ob
    from helpers import flat_unknowns
    te = importlib.import_module("test_energies_gpu")
    out = {}
    for name in NAMES:
        for kind in KINDS:
            for seed in [0] + SEEDS:
                P = te.CASES[name](False)
                o = ob.OracleSolver(P.energy, kind, P.double, P.dims)
                o.set("nIterations", 4); o.set("lIterations", 12)
                if seed:
                    o.set_reduction(1, seed)
                o.init(P.params); c = [o.cost()]
                while o.step(P.params):
                    c.append(o.cost())
                c.append(o.cost())
                out[f"{name}|{kind}|{seed}"] = {"costs": c, "x": flat_unknowns(P).astype(np.float64).tolist()}
                o.close()
    json.dump(out, sys.stdout)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker()
    res = {}
    for variant in ("plain", "fma"):
        env = dict(os.environ)
        if variant == "fma":
            env["OPT_ORACLE_VARIANT"] = "fma"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], capture_output=True, text=True, env=env, cwd=os.path.join(ROOT, "tests"))
        assert r.returncode == 0, r.stderr[-2000:]
        res[variant] = json.loads(r.stdout)
    import numpy as np
    out = {}
    for name in NAMES:
        for kind in KINDS:
            runs, xs = {}, []
            for variant in ("plain", "fma"):
                for seed in [0] + SEEDS:
                    e = res[variant][f"{name}|{kind}|{seed}"]
                    label = ("exact-order " if seed == 0 else f"reference-order seed {seed} ") + variant
                    runs[label] = e["costs"]; xs.append(np.array(e["x"]))
            ref = xs[0]; nrm = max(float(np.linalg.norm(ref)), 1e-300)
            out[f"{name}_{kind}"] = {"runs": runs, "x_spread": max(float(np.linalg.norm(x - ref)) / nrm for x in xs)}
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    for k, e in out.items():
        n = min(len(c) for c in e["runs"].values()); a = e["runs"]["exact-order plain"]; s = max(abs(a[0]), 1e-300)
        d = [max(c[i] for c in e["runs"].values()) - min(c[i] for c in e["runs"].values()) for i in range(n)]
        print(k, "cost spread per step", ["%.1e" % (d[i] / max(abs(a[i]), 1e-7 * s)) for i in range(n)], "x spread %.1e" % e["x_spread"])


if __name__ == "__main__":
    main()
