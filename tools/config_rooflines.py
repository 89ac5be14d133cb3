#!/usr/bin/env python
"""A `roofline` object per BASELINE config: the dominant kernel(s) of configs 1-4 with their hipEvent time (tools/bench_configs.py), the HBM bytes
rocprofv3's PMC passes measured for them (tools/profile_config.sh: 2 x FETCH_SIZE + WRITE_SIZE, per launch) and the bytes the kernel HAS to move (its own model, DESIGN.md
section 3), against two ceilings: HBM's 8 TB/s and -- for working sets that sit in the 256 MiB Infinity Cache -- the copy rate THIS box measured on a 128 MiB working
set (bench.py `box.copy_gbs_cache_resident_128MiB`).

    python tools/config_rooflines.py gpurun_out/<tag> profiles/<tag>_configs.json

Reads <src>/configs.json (one JSON line per config), <src>/bench.json (the box object) and <src>/<cfg>/{kt,pmc_fetch,pmc_write} where a config was profiled.
"""
import collections
import csv
import glob
import json
import os
import sys

HBM_PEAK = 8000.0
# dominant kernel per config: (name in kernel_avg_us, substring of the rocprof kernel name, bytes the kernel has to move per launch, what that model is)
MODELS = {
    "config1": [("PCGIteration", "march_pcgIter", 65.0 * 256 * 256, "poisson marching iteration: 65 B/px (DESIGN 3.5)")],
    "poisson_image_editing 2048": [("PCGIteration", "march_pcgIter", 65.0 * 2048 * 2048, "poisson marching iteration: 65 B/px (DESIGN 3.5)")],
    "config2": [("PCGIteration", "iw_pcgIter2", 53.0 * 2048 * 2048, "image_warping iteration: 53 B/px (DESIGN 3.1)")],
    "config3": [("PCGIteration", "sfs_pcgMarch", 126.0 * 1024 * 1024, "SFS double LM iteration: ~126 B/px (DESIGN 3.5); working set 132 MB: Infinity-Cache-resident")],
    "config4": [("PCGStep1", "arap_applyEll", 180.0 * 500556, "ARAP plane gather (round 6): own planes 80 + r, M 48 + ELL ids 28 + A p 24 = 180 B per vertex = 90 MB (gathered plane entries are the same arrays: L2 / Infinity-Cache hits)"),
                ("PCGStep2+PCGStep3", "arap_flatStepPlanes", 224.0 * 500556, "ARAP flat PCGStep2 + PCGStep3 + dynamic planes: delta, p, r, A p, M in (120) + delta, r, p (72) + D0, D1 (32) out = 224 B per vertex = 112 MB")],
}


# The on-chip shape_from_shading solve (sfs_onchip.h) moves almost nothing through HBM: it is bound by VALU issue.  Instructions of one marching trip (a wave, one held
# row) from the ISA of the shipped variants ((instr(R = 10) - instr(R = 6)) / 4, double LM): 343 in all, 199 of them VALU; a wave64 VALU instruction occupies its SIMD for
# 4 cycles.  Per PCG iteration a SIMD issues (waves per SIMD) x (R + 4) trips.
SFS_TRIP_VALU, SFS_TRIP_ALL, CLOCK_HZ = 199.0, 343.0, 2.4e9


def sfs_onchip_roofline(r):
    us = r["kernel_avg_us"].get("PCGSolveOnChip")
    d = (r.get("plan") or {}).get("describe") or {}
    if not us or "onchip_rows_per_wave" not in d:
        return None
    R, waves = int(d["onchip_rows_per_wave"]), int(d["waves_per_workgroup"])
    L = 10      # lIterations of the config (an LM early-out can end a launch sooner: the bound below is then pessimistic for the kernel)
    trips = (waves // 4) * (R + 4)
    valu_us = L * trips * SFS_TRIP_VALU * 4 / CLOCK_HZ * 1e6
    return {"kernel": "PCGSolveOnChip = sfs_onchipPcg (one launch per linear solve of up to 10 PCG iterations)", "avg_us": us, "us_per_pcg_iteration": us / L, "bound": "valu-issue",
            "model": f"{waves // 4} wave(s) per SIMD x {R + 4} marching trips per iteration x {SFS_TRIP_VALU:.0f} VALU instructions per trip x 4 cycles at {CLOCK_HZ / 1e9:.1f} GHz",
            "valu_issue_us_per_launch": valu_us, "frac": valu_us / us, "unit": "fraction of the launch during which the SIMDs issue the marching trips' VALU instructions",
            "note": "HBM sees only the constants' re-reads (cache hits) and the ring words; all " + f"{SFS_TRIP_ALL:.0f}" + " instructions of a trip (scalar ones included) at 4 cycles would be "
                    f"{L * trips * SFS_TRIP_ALL * 4 / CLOCK_HZ * 1e6:.0f} us", "variant": {"rows_per_wave": R, "waves_per_workgroup": waves}}


def counters(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def traffic(src, cfgdir, needle):
    f = glob.glob(os.path.join(src, cfgdir, "pmc_fetch", "*counter_collection.csv"))
    w = glob.glob(os.path.join(src, cfgdir, "pmc_write", "*counter_collection.csv"))
    if not f or not w:
        return None
    F, Wr = counters(f[0]), counters(w[0])
    k = [n for n in F if needle in n]
    if not k:
        return None
    rd = sum(2 * F[n] * 1024 for n in k) / len(k)      # gfx950: FETCH_SIZE counts half of the streamed bytes, unit KiB (MI355X_MICROARCH.md)
    wr = sum(Wr.get(n, 0.0) * 1024 for n in k) / len(k)
    return {"read": rd, "write": wr, "total": rd + wr}


def main(src, out):
    box = None
    try:
        box = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]).get("box")
    except (OSError, ValueError, IndexError):
        pass
    cache_peak = (box or {}).get("copy_gbs_cache_resident_128MiB")
    rows = []
    for ln in open(os.path.join(src, "configs.json")):
        ln = ln.strip()
        if not ln.startswith("{"):
            continue
        r = json.loads(ln)
        key = next((k for k in MODELS if k in r["config"]), None)
        if key:
            r["roofline"] = []
            if key == "config1" and r["kernel_avg_us"].get("PCGSolveOnChip"):      # 256^2: the whole linear solve is one persistent launch (stencil_onchip.h); nothing streams
                us = r["kernel_avg_us"]["PCGSolveOnChip"]
                r["roofline"].append({"kernel": "PCGSolveOnChip = march_onchipPcg (one launch per linear solve of 10 PCG iterations)", "avg_us": us, "us_per_pcg_iteration": us / 10,
                                      "bound": "latency", "model": "one grid-wide wait per iteration (tagged words through the memory side: ~3.5 us) + the stencil on 2 rows per wave; 65 k pixels x 65 B = 4 MB of state, all of it in registers",
                                      "note": "no bandwidth or issue ceiling applies at this size: the launch-per-iteration kernel it replaces took 11.4 us per iteration, all launch latency", "plan": (r.get("plan") or {}).get("describe")})
            if key == "config3":
                o = sfs_onchip_roofline(r)
                if o:
                    t = traffic(src, "pmc_config3", "sfs_onchipPcg")
                    o.update({"traffic": t["total"] if t else None, "traffic_read_write": t})      # HBM bytes per LAUNCH (a whole linear solve)
                    r["roofline"].append(o)
            for kname, needle, model, what in MODELS[key]:
                us = r["kernel_avg_us"].get(kname)
                if not us:
                    continue
                t = traffic(src, "pmc_" + key.split()[0], needle)
                ach = model / (us * 1e-6) / 1e9
                o = {"kernel": f"{kname} = {needle}", "avg_us": us, "bound": "hbm", "model_bytes_per_launch": model, "model": what, "achieved": ach, "unit": "GB/s",
                     "peak": HBM_PEAK, "frac": ach / HBM_PEAK, "traffic": t["total"] if t else None, "traffic_read_write": t,
                     "hbm_achieved": t["total"] / (us * 1e-6) / 1e9 if t else None}
                if box and box.get("copy_gbs_float4_nontemporal"):      # round 6: what a float4 copy kernel reaches on THIS box (OptAmd_MeasureCopyBandwidth)
                    best = max(box.get("copy_gbs_float4") or 0.0, box["copy_gbs_float4_nontemporal"])
                    o.update({"box_copy_float4_gbs": best, "frac_of_box_copy_float4": (o["hbm_achieved"] or ach) / best})
                if "Infinity-Cache-resident" in what and cache_peak:
                    o.update({"cache_resident_peak_measured": cache_peak, "frac_of_cache_resident_peak": ach / cache_peak,
                              "note": "the working set sits in the 256 MiB Infinity Cache: HBM sees only part of the bytes (traffic < model); the honest ceiling is the box's own copy rate on a 128 MiB working set"})
                r["roofline"].append(o)
        rows.append(r)
    json.dump({"box": box, "configs": rows}, open(out, "w"), indent=1)
    for r in rows:
        for o in r.get("roofline", []):
            if o.get("bound") == "latency":
                print(f"{r['config'][:60]:60s} {o['kernel'][:40]:40s} {o['avg_us']:8.1f} us  = {o['us_per_pcg_iteration']:.1f} us per PCG iteration (latency-bound)")
                continue
            if o.get("bound") == "valu-issue":
                print(f"{r['config'][:60]:60s} {o['kernel'][:40]:40s} {o['avg_us']:8.1f} us  VALU issue {o['valu_issue_us_per_launch']:.0f} us = {o['frac']:.2f} of the launch")
                continue
            print(f"{r['config'][:60]:60s} {o['kernel']:40s} {o['avg_us']:8.1f} us  model {o['achieved']:7.0f} GB/s = {o['frac']:.2f} of HBM peak"
                  + (f", {o['frac_of_cache_resident_peak']:.2f} of the measured cache-resident copy rate" if "frac_of_cache_resident_peak" in o else "")
                  + (f"; PMC traffic {o['traffic'] / 1e6:.0f} MB" if o["traffic"] else ""))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
