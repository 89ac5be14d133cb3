#!/usr/bin/env python
"""The HIP side's OWN ensemble of legal runs at long horizons, next to the frozen ensemble of the reference's arithmetic.

A run of the reference is one draw from a distribution: its dot products are summed by float atomics in an order the hardware does not define
(tests/golden/make_reference_order_spread.py freezes seeded draws of the oracle's reference-order mode).  The HIP loops are deterministic for a given launch
geometry, but every geometry is another legal summation order: the rows a marching workgroup takes (OPT_AMD_ITER_ROWS) changes which pixels share a partial sum
and the order in which the partials are added.  This tool runs image_warping 2048^2 float (the `horizon` family of tests/golden/make_horizon_costs.py), ONE
Gauss-Newton step of L PCG iterations, for

    loops      ref-order (OPT_AMD_ONEKERNEL=0: the reference's three kernels per iteration)  and  r-free (one launch per iteration: the benchmarked kernel)
    geometries OPT_AMD_ITER_ROWS in ROWS (+ the default: one co-resident wave of workgroups)

and compares the two ensembles as DISTRIBUTIONS, anchored on the median of the reference-order runs (not on the long-double oracle, which is no run of the reference):
quantiles of the signed relative deviation, the fraction of runs within the 1e-5 contract of that median, and the two-sample Kolmogorov-Smirnov statistic.

    python tools/horizon_ensemble.py [--out gpurun_out/ensemble] [--horizons 20 50 100 400] [--traces 50]

Writes <out>.json / <out>.md; --traces L also saves the per-iteration scalars (OptAmd_PlanGetTrace) of both loops at that horizon as <out>_trace_<loop>_<L>.npy
(tools/round5/l50_bisect.py puts them next to the oracle's).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import reference_spread as rs      # noqa: E402

ROWS = [0, 5, 6, 7, 9, 11, 13, 16, 19, 23, 29, 37, 47, 61, 79, 98, 131]      # 0: the default geometry
LOOPS = {"ref-order": {"OPT_AMD_ONEKERNEL": "0", "OPT_AMD_ONCHIP": "0"}, "r-free": {"OPT_AMD_ONCHIP": "0"}}
SIZE = 2048


def reference_ensemble(L):
    """Signed costs of the frozen reference-order runs (every seed, plain / fma build, banded / raster scatter) at horizon L."""
    return [c for lab, c in rs.legal_runs(f"horizon_{SIZE}_float_{L}") if lab.startswith("reference-order")]


def hip_run(L, loop, rows, trace=False):
    import torch
    from opt_amd import api, workloads as wl
    env = dict(LOOPS[loop])
    if rows:
        env["OPT_AMD_ITER_ROWS"] = str(rows)
    old = {k: os.environ.get(k) for k in list(env) + ["OPT_AMD_ITER_ROWS"]}
    os.environ.pop("OPT_AMD_ITER_ROWS", None)
    os.environ.update(env)
    try:
        P = wl.image_warping(SIZE, SIZE)
        dev = api.to_device(P)
        s = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", P.dims)
        s.set_parameter("nIterations", 1); s.set_parameter("lIterations", L)
        if trace:
            s.enable_trace(True)
        s.init(dev); s.step(dev)
        torch.cuda.synchronize()
        c = s.cost()
        t = s.trace() if trace else None
        s.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return c, t


def compare(hip, ref):
    """Distribution summary of two samples of costs, anchored on the median of `ref`."""
    from scipy import stats
    med = float(np.median(ref))
    dh, dr = (np.asarray(hip) - med) / med, (np.asarray(ref) - med) / med
    q = lambda d: [float(np.quantile(d, p)) for p in (0.1, 0.25, 0.5, 0.75, 0.9)]
    ks = stats.ks_2samp(dh, dr)
    return {"reference_median": med, "n_hip": len(hip), "n_ref": len(ref), "hip_quantiles_10_25_50_75_90": q(dh), "ref_quantiles_10_25_50_75_90": q(dr),
            "hip_within_1e-5_of_ref_median": float(np.mean(np.abs(dh) <= 1e-5)), "ref_within_1e-5_of_ref_median": float(np.mean(np.abs(dr) <= 1e-5)),
            "hip_max_abs": float(np.max(np.abs(dh))), "ref_max_abs": float(np.max(np.abs(dr))), "ks_statistic": float(ks.statistic), "ks_pvalue": float(ks.pvalue)}


def markdown(res):
    f = lambda v: f"{v:+.2e}"
    out = ["# HIP ensemble against the reference-order ensemble (image_warping 2048^2 float, one Gauss-Newton step of L PCG iterations)", "",
           "Signed relative deviation of the cost from the MEDIAN of the frozen reference-order runs.  HIP ensemble: two loops x launch geometries (tools/horizon_ensemble.py).", "",
           "| L | runs HIP / ref | HIP quantiles 10 / 50 / 90 % | ref quantiles 10 / 50 / 90 % | within 1e-5 of the ref median: HIP / ref | max abs: HIP / ref | KS statistic | KS p |", "|---|---|---|---|---|---|---|---|"]
    for L, r in sorted(res.items(), key=lambda kv: int(kv[0])):
        c = r["compare"]
        hq, rq = c["hip_quantiles_10_25_50_75_90"], c["ref_quantiles_10_25_50_75_90"]
        out.append(f"| {L} | {c['n_hip']} / {c['n_ref']} | {f(hq[0])} / {f(hq[2])} / {f(hq[4])} | {f(rq[0])} / {f(rq[2])} / {f(rq[4])} | {c['hip_within_1e-5_of_ref_median']:.2f} / {c['ref_within_1e-5_of_ref_median']:.2f} | "
                   f"{c['hip_max_abs']:.2e} / {c['ref_max_abs']:.2e} | {c['ks_statistic']:.2f} | {c['ks_pvalue']:.3f} |")
    out += ["", "Individual HIP runs (signed deviation from the reference median; rows = OPT_AMD_ITER_ROWS, 0 = default geometry):", ""]
    for L, r in sorted(res.items(), key=lambda kv: int(kv[0])):
        med = r["compare"]["reference_median"]
        for loop in LOOPS:
            out.append(f"* L = {L}, {loop}: " + ", ".join(f"{rows}: {(c - med) / med:+.1e}" for rows, c in r["hip"][loop]))
    return "\n".join(out) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ensemble"))
    ap.add_argument("--horizons", type=int, nargs="+", default=[20, 50, 100, 400])
    ap.add_argument("--traces", type=int, nargs="*", default=[])
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    res = {}
    for L in a.horizons:
        ref = reference_ensemble(L)
        hip = {loop: [(rows, hip_run(L, loop, rows)[0]) for rows in ROWS] for loop in LOOPS}
        allh = [c for loop in LOOPS for _, c in hip[loop]]
        res[str(L)] = {"hip": hip, "reference_order": ref, "exact_order_anchor": rs.anchor(f"horizon_{SIZE}_float_{L}"), "compare": compare(allh, ref) if len(ref) >= 2 else None}
        print(L, json.dumps(res[str(L)]["compare"]), flush=True)
    for L in a.traces:
        for loop in LOOPS:
            c, t = hip_run(L, loop, 0, trace=True)
            np.save(f"{a.out}_trace_{loop}_{L}.npy", t)
            print("trace", loop, L, c, flush=True)
    json.dump(res, open(a.out + ".json", "w"), indent=1)
    open(a.out + ".md", "w").write(markdown(res))
    print(markdown(res))


if __name__ == "__main__":
    main()
