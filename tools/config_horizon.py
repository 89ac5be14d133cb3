#!/usr/bin/env python
"""BASELINE configs 3 and 4 at their OWN horizons against the frozen oracle cost histories (tests/golden/config_costs.json, make_config_costs.py):

  config 3  shape_from_shading 1024^2 double LM 60 x 10        config 4  arap_mesh_deformation 708 x 707 float GN 20 x 100

Both HIP paths -- the default one (on-chip / fused loops) and the reference-ordered loop (Opt_SetSolverParameter amd_reference_order = 1) -- stepped side by side with the
frozen oracle: relative cost error after every outer step, (LM) the trust-region radius, where the run first leaves the contract and at what rate the distance grows.

    python tools/config_horizon.py [--out gpurun_out/config_horizon.json] [--configs 3 4]

tests/test_config_horizon_gpu.py asserts the bars derived from this table (tests/golden/config_horizon_bars.json, written with --freeze).
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "config_costs.json")
BARS = os.path.join(ROOT, "tests", "golden", "config_horizon_bars.json")

CASES = {
    3: ("config3_sfs_1024_double_LM_60x10", lambda wl: wl.shape_from_shading(1024, 1024, double=True, holes=True), "LMGPU", 60, 10, 1e-12),
    4: ("config4_arap_708x707_float_GN_20x100", lambda wl: wl.arap_mesh_deformation(708, 707, perturb=0.01), "gaussNewtonGPU", 20, 100, 1e-5),
}


def run(cfg, reference_order):
    import torch
    from opt_amd import api, workloads as wl
    key, make, kind, n_it, l_it, _ = CASES[cfg]
    P = make(wl)
    dev = api.to_device(P)
    s = api.Solver(api.energy_file(P.energy), kind, P.dims, double=P.double)
    s.set_parameter("nIterations", n_it); s.set_parameter("lIterations", l_it)
    if reference_order:
        s.set_parameter("amd_reference_order", 1)
    s.init(dev)
    costs, radii = [s.cost()], [s.trust_region_radius() if kind == "LMGPU" else 0.0]
    while s.step(dev):
        costs.append(s.cost()); radii.append(s.trust_region_radius() if kind == "LMGPU" else 0.0)
    torch.cuda.synchronize()
    st = s.on_chip_status()
    s.close()
    return costs, radii, st


def compare(cfg, G):
    key, _, kind, n_it, l_it, contract = CASES[cfg]
    g = G[key]
    out = {"key": key, "contract": contract, "oracle_steps": g["steps_taken"], "paths": {}}
    for name, ref in (("default", False), ("reference_order", True)):
        costs, radii, st = run(cfg, ref)
        n = min(len(costs), len(g["costs"]))
        rel = [abs(a - b) / abs(b) for a, b in zip(costs[:n], g["costs"][:n])]
        rad = [abs(a - b) / abs(b) if b else 0.0 for a, b in zip(radii[:n], g.get("radii", [0.0] * n)[:n])]
        first = next((i for i, e in enumerate(rel) if e > contract), None)
        # growth rate per outer step over the stretch where the distance is above round-off and still growing (geometric mean of successive ratios)
        pts = [(i, e) for i, e in enumerate(rel) if e > 0]
        rate = None
        if len(pts) >= 4:
            i0, e0 = pts[1]; i1, e1 = max(pts, key=lambda p: p[1])
            if i1 > i0 and e1 > e0:
                rate = math.exp((math.log(e1) - math.log(e0)) / (i1 - i0))
        out["paths"][name] = {"steps": len(costs) - 1, "same_step_count_as_oracle": len(costs) == len(g["costs"]), "on_chip_status": st, "rel_err_per_step": rel, "radius_rel_err_per_step": rad if kind == "LMGPU" else None,
                              "max_rel_err": max(rel), "final_rel_err": rel[-1], "first_step_outside_contract": first, "growth_per_step_until_max": rate,
                              "final_cost": costs[-1], "oracle_final_cost": g["costs"][n - 1]}
    if cfg == 4 and "config4_arap_708x707_double_GN_20x100" in G:      # the rounding-free yardstick of the float trajectory: the oracle's own float run against its double run
        d = G["config4_arap_708x707_double_GN_20x100"]["costs"]
        out["oracle_float_vs_double_per_step"] = [abs(a - b) / abs(b) for a, b in zip(g["costs"], d)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--configs", type=int, nargs="+", default=[3, 4])
    ap.add_argument("--from-json", default=None, help="take the measurements from an earlier --out file (a run on the GPU box) instead of running: --from-json X --freeze")
    ap.add_argument("--freeze", action="store_true", help="write tests/golden/config_horizon_bars.json: per path and step, max(contract, 10 x measured)")
    args = ap.parse_args()
    G = json.load(open(GOLD))
    res = json.load(open(args.from_json)) if args.from_json else {}
    for c in ([] if args.from_json else args.configs):
        if CASES[c][0] not in G:
            print(f"config {c}: not frozen yet", file=sys.stderr)
            continue
        res[str(c)] = compare(c, G)
        r = res[str(c)]
        for name, p in r["paths"].items():
            e = p["rel_err_per_step"]
            print(f"config {c} {name:16s}: steps {p['steps']} (oracle {r['oracle_steps']}), on_chip_status {p['on_chip_status']}, rel err step 1 {e[1]:.1e}, step 5 {e[min(5, len(e) - 1)]:.1e}, step 10 {e[min(10, len(e) - 1)]:.1e}, "
                  f"step 20 {e[min(20, len(e) - 1)]:.1e}, final {e[-1]:.1e}, max {p['max_rel_err']:.1e}, first outside {r['contract']:.0e}: step {p['first_step_outside_contract']}, growth/step {p['growth_per_step_until_max']}")
        if "oracle_float_vs_double_per_step" in r:
            e = r["oracle_float_vs_double_per_step"]
            print(f"config {c} oracle float vs double: step 1 {e[1]:.1e}, step 5 {e[5]:.1e}, step 10 {e[10]:.1e}, final {e[-1]:.1e}")
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    if args.freeze:
        bars = json.load(open(BARS)) if os.path.exists(BARS) else {}
        for c, r in res.items():
            for name, p in r["paths"].items():
                def up(v):
                    e = math.floor(math.log10(v)); return math.ceil(v / 10 ** e) * 10 ** e
                bars[f"{r['key']}|{name}"] = {"contract": r["contract"], "bar_per_step": [max(r["contract"], up(10 * e)) if e > 0 else r["contract"] for e in p["rel_err_per_step"]],
                                              "measured_per_step": p["rel_err_per_step"], "steps": p["steps"], "first_step_outside_contract": p["first_step_outside_contract"],
                                              "growth_per_step_until_max": p["growth_per_step_until_max"]}
        json.dump(bars, open(BARS, "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
