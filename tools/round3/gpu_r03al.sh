#!/bin/bash
# ARAP symmetric-graph path: two-kernel PCG iteration (flat pass writing the records + gather with the expansion sums)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03al
timeout 500 python -m pytest tests/test_energies_gpu.py tests/test_golden_gpu.py -m gpu -q -k "arap" 2>&1 | tail -6 > gpurun_out/r03al/pytest.log; cat gpurun_out/r03al/pytest.log
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
run OPT_AMD_ARAP_ITER=0
run OPT_AMD_ARAP_SYM=1
run OPT_AMD_ARAP_ITER=0
run OPT_AMD_ARAP_SYM=0
} 2>&1 | tee gpurun_out/r03al/config4_iter.txt
