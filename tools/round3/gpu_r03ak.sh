#!/bin/bash
# ARAP symmetric-graph path: XCD-aware vertex order (L2 misses), checksum with one atomic per workgroup
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03ak
timeout 500 python -m pytest tests/test_energies_gpu.py -m gpu -q -k "arap" 2>&1 | tail -6 > gpurun_out/r03ak/pytest.log; cat gpurun_out/r03ak/pytest.log
show='
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l); k = d["kernel_avg_us"]
        print("wall %.1f ms  cost %.9g  " % (d["wall_s"] * 1e3, d["cost_final"]), {n: round(k[n], 1) for n in k if "Step" in n or "EdgeLists" in n})
'
run() { echo "== $*"; env "$@" OPT_AMD_CONFIG="config4" timeout 200 python tools/bench_configs.py 2>/dev/null | python -c "$show"; }
{
run OPT_AMD_ARAP_SYM_XCD=1
run OPT_AMD_ARAP_SYM_XCD=0
run OPT_AMD_ARAP_SYM_XCD=1 OPT_AMD_ARAP_VGRID=1536
run OPT_AMD_ARAP_SYM_XCD=1 OPT_AMD_ARAP_VGRID=2048
run OPT_AMD_ARAP_SYM_XCD=1 OPT_AMD_ARAP_SYM_LANES=4
run OPT_AMD_ARAP_SYM_XCD=1 OPT_AMD_ARAP_SYM_BATCH=1
run OPT_AMD_ARAP_SYM_XCD=1
} 2>&1 | tee gpurun_out/r03ak/config4_xcd.txt
