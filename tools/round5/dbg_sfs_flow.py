"""round 5: where an outer step of the shape_from_shading flow goes (640 x 480 double LM 60 x 10, and config 3)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from opt_amd import api, workloads as wl
import torch

for (W, H, steps) in [(640, 480, 60), (1024, 1024, 60)]:
    P = wl.shape_from_shading(W, H, double=True, seed=1, holes=True)
    for flag in ("0", "1"):
        os.environ["OPT_AMD_ONCHIP"] = flag
        for timing in (False, True):
            g = api.Solver(api.energy_file(P.energy), "LMGPU", P.dims, double=True, timing=timing)
            g.set_parameter("nIterations", steps); g.set_parameter("lIterations", 10)
            dev = api.to_device(P)
            g.solve(dev)      # warm-up
            dev = api.to_device(P)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g.solve(dev)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if timing:
                t = g.kernel_timings()
                tot = sum(v[1] for k, v in t.items() if k != "overall")
                print(f"   kernels: sum {tot:.2f} ms; " + ", ".join(f"{k} {v[0]}x{1e3 * v[1] / max(v[0], 1):.1f}us" for k, v in t.items()), flush=True)
            else:
                print(f"{W}x{H} f64 LM 60x10 onchip={flag}: wall {dt * 1e3:.2f} ms, final cost {g.cost():.10g}", flush=True)
            g.close()

# step by step against the frozen oracle trajectory of the same flow (tests/golden/sfs_flow_640x480_double_lm_60x10_oracle.json; oracle/ itself does not run here)
import json
gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "sfs_flow_640x480_double_lm_60x10_oracle.json")))
P = wl.shape_from_shading(640, 480, double=True, seed=1, holes=True)
for flag in ("0", "1"):
    os.environ["OPT_AMD_ONCHIP"] = flag
    g = api.Solver(api.energy_file(P.energy), "LMGPU", P.dims, double=True)
    g.set_parameter("nIterations", 60); g.set_parameter("lIterations", 10)
    dev = api.to_device(P)
    g.init(dev); costs = [g.cost()]; radii = [g.trust_region_radius()]
    while g.step(dev):
        costs.append(g.cost()); radii.append(g.trust_region_radius())
    costs.append(g.cost())
    g.close()
    rel = [abs(a - b) / abs(b) for a, b in zip(costs, gold["costs"])]
    first = next((i for i, e in enumerate(rel) if e > 1e-9), None)
    print(f"onchip={flag}: {len(costs)} costs (oracle {len(gold['costs'])}); first step more than 1e-9 from the oracle: {first}; rel by step: " + " ".join(f"{e:.1e}" for e in rel), flush=True)
    if first is not None:
        print("   radii around it (hip / oracle):", [(radii[i], gold["radii"][i]) for i in range(max(0, first - 2), min(len(radii), len(gold["radii"]), first + 2))], flush=True)
