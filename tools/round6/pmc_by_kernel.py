#!/usr/bin/env python
"""Mean counter value per dispatch, per (kernel, counter), over the passes of tools/round6/arap_pmc.sh (counter rows of one dispatch are summed over their dimensions first)."""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"<.*", "", r["Kernel_Name"]).replace("void optamd::(anonymous namespace)::", "").replace("optamd::(anonymous namespace)::", "")
        agg[n][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"<.*", "", r["Kernel_Name"]).replace("void optamd::(anonymous namespace)::", "").replace("optamd::(anonymous namespace)::", "")
        dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k in sorted(agg, key=lambda k: -sum(dur.get(k, [0]))):
    d = dur.get(k, [])
    print(f"{k}   launches {len(d)}  mean {sum(d) / max(1, len(d)):.2f} us  total {sum(d) * 1e-3:.2f} ms")
    for c in sorted(agg[k]):
        v = list(agg[k][c].values())
        print("   %-36s %18.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
