#!/bin/bash
# ARAP symmetric-graph path: slots / records requested in batches
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03ag
timeout 500 python -m pytest tests/test_energies_gpu.py -m gpu -q -k "arap" 2>&1 | tail -6 > gpurun_out/r03ag/pytest.log; cat gpurun_out/r03ag/pytest.log
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
for b in 1 2 3 4; do run OPT_AMD_ARAP_SYM_BATCH=$b; done
run OPT_AMD_ARAP_SYM_BATCH=3 OPT_AMD_ARAP_VGRID=1536
run OPT_AMD_ARAP_SYM_BATCH=3 OPT_AMD_ARAP_VGRID=768
run OPT_AMD_ARAP_SYM_BATCH=2 OPT_AMD_ARAP_SYM_LANES=4
run OPT_AMD_ARAP_SYM_BATCH=2 OPT_AMD_ARAP_SYM_LANES=4 OPT_AMD_ARAP_VGRID=1536
run OPT_AMD_ARAP_SYM_BATCH=4 OPT_AMD_ARAP_SYM_LANES=1
run OPT_AMD_ARAP_SYM_BATCH=3
} 2>&1 | tee gpurun_out/r03ag/config4_batch.txt
