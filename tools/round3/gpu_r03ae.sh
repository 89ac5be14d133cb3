#!/bin/bash
# ARAP symmetric-graph path: p from the records vs from the solver's vector
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03ae
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
run OPT_AMD_ARAP_SYM_PVEC=1
run OPT_AMD_ARAP_SYM_PVEC=1 OPT_AMD_ARAP_SYM_LANES=4
run OPT_AMD_ARAP_SYM_PVEC=1 OPT_AMD_ARAP_SYM_LANES=4 OPT_AMD_ARAP_VGRID=1536
run OPT_AMD_ARAP_STEP3_VEC=1
run OPT_AMD_ARAP_SYM=1
run OPT_AMD_ARAP_SYM_PVEC=1
} 2>&1 | tee gpurun_out/r03ae/config4_pvec.txt
