#!/usr/bin/env python
"""Aggregate tools/pmc_passes.sh output: mean counter value per (kernel, counter)."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = "K1f" if "iw_applyJTJ" in n and ", true>(" in n else "K2" if "k_step2<" in n else "ITER" if "iw_pcgIter" in n else None
        if key:
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print("   %-34s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
