#!/bin/bash
# ARAP record path with the XCD-aware order and the two-kernel iteration: lanes x grid x batch again
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03aq
show='
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l); k = d["kernel_avg_us"]
        print("wall %.1f ms  cost %.9g  " % (d["wall_s"] * 1e3, d["cost_final"]), {n: round(k[n], 1) for n in k if "Step" in n})
'
run() { echo "== $*"; env "$@" OPT_AMD_CONFIG="config4" timeout 200 python tools/bench_configs.py 2>/dev/null | python -c "$show"; }
{
run OPT_AMD_ARAP_SYM=1
for l in 2 4; do for g in 1024 1536 2048; do for b in 1 4; do run OPT_AMD_ARAP_SYM_LANES=$l OPT_AMD_ARAP_VGRID=$g OPT_AMD_ARAP_SYM_BATCH=$b; done; done; done
run OPT_AMD_ARAP_SYM=1
} 2>&1 | tee gpurun_out/r03aq/config4_retune.txt
