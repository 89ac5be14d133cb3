#!/bin/bash
# SQ counters of the SFS iteration kernels (marching vs tiled): where do the waves spend their cycles?
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OPT_AMD_CONFIG=config3 OPT_AMD_NO_TIMING_RUN=1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -f csv -d $O/sq_march -o p -- python tools/bench_configs.py > $O/sq_march.log 2>&1
OPT_AMD_SFS_MARCH=0 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -f csv -d $O/sq_tiled -o p -- python tools/bench_configs.py > $O/sq_tiled.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("sq_march", "sq_tiled"):
    files = glob.glob(f"gpurun_out/r03h/{tag}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:4]:
        n = max(cnt[k], 1)
        print(tag, k, "launches", n, {c: round(x / n) for c, x in v.items()})
PY
