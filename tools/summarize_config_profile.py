#!/usr/bin/env python
"""Per-kernel table (calls, average time, HBM bytes per launch, GB/s) of a tools/profile_config.sh run.

    python tools/summarize_config_profile.py gpurun_out/<tag> profiles/<tag>.md "<title>"

FETCH_SIZE is doubled (gfx950 reports half of the streamed bytes, MI355X_MICROARCH.md; verified in profiles/r01*_summary.md),
WRITE_SIZE is used as reported.  Kernels whose working set sits in the 256 MB Infinity Cache show little HBM traffic: their
GB/s column is then far below what they move through the cache -- the time column is what matters there.
"""
import collections
import csv
import glob
import os
import sys


def mean_counter(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def short(name):
    n = name.replace("optamd::(anonymous namespace)::", "").replace("optamd::", "").replace("void ", "")
    return n.split("(")[0].strip()


def main(src, out, title):
    rows = list(csv.DictReader(open(glob.glob(os.path.join(src, "kt", "*kernel_stats.csv"))[0])))
    fetch = mean_counter(glob.glob(os.path.join(src, "pmc_fetch", "*counter_collection.csv"))[0])
    write = mean_counter(glob.glob(os.path.join(src, "pmc_write", "*counter_collection.csv"))[0])
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = ["# " + title, "", "`rocprofv3 --kernel-trace --stats` + separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of `tools/bench_configs.py` "
             "restricted to this config (warm-up solve + timed solve + per-kernel-timing solve).", "",
             "| kernel | calls | avg us | % of GPU time | HBM read MB (2 x FETCH_SIZE) | HBM write MB | HBM GB/s |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        n = r["Name"]
        avg = float(r["AverageNs"]) / 1e3
        rd, wr = 2 * fetch.get(n, 0.0) * 1024 / 1e6, write.get(n, 0.0) * 1024 / 1e6
        pct = 100 * float(r["TotalDurationNs"]) / total
        if pct < 0.3:
            continue
        lines.append(f"| {short(n)} | {r['Calls']} | {avg:.1f} | {pct:.1f} | {rd:.1f} | {wr:.1f} | {(rd + wr) / 1e3 / (avg * 1e-6):.0f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else os.path.basename(sys.argv[2]))
