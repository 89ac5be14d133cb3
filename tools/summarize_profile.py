#!/usr/bin/env python
"""Fold a rocprofv3 run directory (gpurun_out/<tag>/{kt,pmc_fetch,pmc_write}) into profiles/<tag>_*.

    python tools/summarize_profile.py gpurun_out/r01 profiles/r01

Writes <out>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim), <out>_bench.json (the bench line of
the same command set) and <out>_summary.md: per-kernel average duration, FETCH_SIZE / WRITE_SIZE per launch with
the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced
stream -- verified here on kernels with a known byte count), and the resulting HBM traffic vs algorithmic bytes.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def mean_counter(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def short(name):
    n = name.replace("optamd::(anonymous namespace)::", "").replace("optamd::", "").replace("void ", "")
    return n.split("(")[0].strip()


def main(src, out):
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    stats = glob.glob(os.path.join(src, "kt", "*kernel_stats.csv"))[0]
    shutil.copy(stats, out + "_kernel_stats.csv")
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), out + "_bench.json")
    rows = list(csv.DictReader(open(stats)))
    fetch = mean_counter(glob.glob(os.path.join(src, "pmc_fetch", "*counter_collection.csv"))[0])
    write = mean_counter(glob.glob(os.path.join(src, "pmc_write", "*counter_collection.csv"))[0])
    lines = ["# rocprofv3 summary: " + os.path.basename(out), "",
             "Source: `rocprofv3 --kernel-trace --stats -f csv` and two separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of",
             "`python bench.py --steps 1 --warmup 0 --liters 50 --no-cpu-baseline` (image_warping 4096x4096 float).",
             "FETCH_SIZE/WRITE_SIZE are in KiB per launch; on gfx950 FETCH_SIZE reports 1/2 of the streamed bytes (MI355X_MICROARCH.md, HBM section),",
             "so `HBM read` = 2 x FETCH_SIZE; WRITE_SIZE is used as reported (it matches the known byte count of k_step2: 3 vectors = 589,824 KiB).", "",
             "| kernel | calls | avg us | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM read MB (2x) | HBM write MB | GB/s (read+write)/avg |",
             "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r["Name"]
        avg_us = float(r["AverageNs"]) / 1e3
        f = fetch.get(name, (0, 0.0))[1]
        w = write.get(name, (0, 0.0))[1]
        rd_mb, wr_mb = 2 * f * 1024 / 1e6, w * 1024 / 1e6
        gbs = (rd_mb + wr_mb) / 1e3 / (avg_us * 1e-6) if avg_us > 0 else 0
        lines.append(f"| {short(name)} | {r['Calls']} | {avg_us:.1f} | {f:.0f} | {w:.0f} | {rd_mb:.1f} | {wr_mb:.1f} | {gbs:.0f} |")
    # traffic of the dominant kernel for bench.py's roofline.traffic (bench_kernel = the name bench.py times it under).  The iteration
    # kernel exists in two template variants that alternate launch by launch (sweep direction; the paired delta update makes the
    # even launches heavier), so the per-launch figure is the call-weighted mean over all variants.
    groups = {"PCGIteration": [r for r in rows if "iw_pcgIter2" in r["Name"]] or [r for r in rows if "iw_pcgIter" in r["Name"]],
              "PCGStep3+PCGStep1": [r for r in rows if "iw_applyJTJ" in r["Name"] and ", true>(" in r["Name"]]}
    sha = None
    try:
        sha = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1]).get("kernel_src_sha16")
    except Exception:  # noqa
        pass
    if sha is None:      # no bench line yet (tools/round_artifacts.sh folds the counters before it runs the bench): hash the sources of this tree
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        sha = bench.kernel_src_sha16()
    for bench_kernel, rs in groups.items():
        if not rs:
            continue
        calls = sum(int(r["Calls"]) for r in rs)
        f = sum(int(r["Calls"]) * fetch.get(r["Name"], (0, 0.0))[1] for r in rs) / calls
        w = sum(int(r["Calls"]) * write.get(r["Name"], (0, 0.0))[1] for r in rs) / calls
        avg = sum(int(r["Calls"]) * float(r["AverageNs"]) for r in rs) / calls / 1e3
        json.dump({"kernel": " + ".join(sorted(short(r["Name"]) for r in rs)), "bench_kernel": bench_kernel, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
                   "fetch_size_kib": f, "write_size_kib": w, "correction": "2 x FETCH_SIZE (gfx950) + WRITE_SIZE, call-weighted mean over the kernel's variants",
                   "avg_us_rocprof": avg, "launches": calls, "kernel_src_sha16": sha, "workload": "image_warping 4096x4096 float", "source": os.path.basename(out)}, open(out + "_traffic.json", "w"))
        lines += ["", f"dominant kernel ({bench_kernel}): {calls} launches, call-weighted mean {avg:.1f} us, {(2 * f + w) * 1024 / 1e9:.3f} GB of HBM traffic per launch "
                      f"= {(2 * f + w) * 1024 / (4096 * 4096):.1f} B/pixel, {(2 * f + w) * 1024 / 1e3 / avg:.0f} GB/s"]
        break
    if os.path.exists(os.path.join(src, "bench.json")):
        try:
            b = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
            lines += ["", "bench.py line of the same box: value = %.1f %s, roofline = %s" % (b["value"], b["unit"], json.dumps({k: b["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_kernel_ms")} if b.get("roofline") else None))]
        except Exception as e:  # noqa
            lines += ["", f"(bench.json unreadable: {e})"]
    open(out + "_summary.md", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
