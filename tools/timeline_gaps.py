#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV: where a latency-bound solve (LM with host decisions) loses its time.
    python tools/timeline_gaps.py <dir with *kernel_trace.csv> [skip_first_n]
Prints busy / span of the trace's last solve and the idle time grouped by (kernel before the gap -> kernel after it)."""
import csv, glob, os, re, sys
from collections import defaultdict

def short(n):
    n = re.sub(r"\(anonymous namespace\)::|optamd::|void ", "", n)
    return re.sub(r"[<(].*", "", n)

def main():
    d = sys.argv[1]
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
    rows.sort()
    # the timed solve = the last run of kernels after the longest idle stretch in the second half of the trace
    tail = rows[len(rows) // 2:]
    cut = max(range(1, len(tail)), key=lambda i: tail[i][0] - tail[i - 1][1])
    sel = tail[cut:]
    span = sel[-1][1] - sel[0][0]
    busy = sum(e - s for s, e, _ in sel)
    print(f"{os.path.relpath(f)}: last solve {len(sel)} launches, span {span/1e3:.1f} us, busy {busy/1e3:.1f} us ({busy/span:.2f}), idle {(span-busy)/1e3:.1f} us")
    gaps = defaultdict(lambda: [0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(sel, sel[1:]):
        g = gaps[(n0, n1)]; g[0] += 1; g[1] += max(0, s1 - e0)
    print(f"{'gap after -> before':60s} {'count':>6s} {'total us':>10s} {'avg us':>8s}")
    for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"{a + ' -> ' + b:60s} {c:6d} {t/1e3:10.1f} {t/1e3/c:8.2f}")
    per = defaultdict(lambda: [0, 0])
    for s, e, n in sel: per[n][0] += 1; per[n][1] += e - s
    print(f"\n{'kernel':40s} {'count':>6s} {'total us':>10s} {'avg us':>8s}")
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]): print(f"{n:40s} {c:6d} {t/1e3:10.1f} {t/1e3/c:8.2f}")

if __name__ == "__main__":
    main()
