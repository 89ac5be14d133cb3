#!/bin/bash
# ARAP symmetric-graph path (arap_applySym): parity tests + config 4 A/B against the edge-list gather
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03aa
timeout 500 python -m pytest tests/test_energies_gpu.py -m gpu -q -x -k "arap" 2>&1 | tail -15 > gpurun_out/r03aa/pytest.log; cat gpurun_out/r03aa/pytest.log
show='
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l); k = d["kernel_avg_us"]
        print("wall %.1f ms  cost %.9g  " % (d["wall_s"] * 1e3, d["cost_final"]), {n: round(k[n], 1) for n in k})
'
for rep in 1 2; do for m in 1 0; do echo "== config4 sym=$m"; OPT_AMD_ARAP_SYM=$m OPT_AMD_CONFIG="config4" timeout 200 python tools/bench_configs.py 2>/dev/null | python -c "$show"; done; done 2>&1 | tee gpurun_out/r03aa/config4_ab.txt
