#!/bin/bash
# ARAP symmetric-graph path: grid size x lanes per vertex
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03ad
timeout 300 python -m pytest tests/test_energies_gpu.py -m gpu -q -k "arap_symmetric" 2>&1 | tail -4 > gpurun_out/r03ad/pytest.log; cat gpurun_out/r03ad/pytest.log
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
for g in 768 1024 1280 1536; do for l in 2 4; do run OPT_AMD_ARAP_SYM_LANES=$l OPT_AMD_ARAP_VGRID=$g; done; done
run OPT_AMD_ARAP_SYM_LANES=1 OPT_AMD_ARAP_VGRID=1024
run OPT_AMD_ARAP_SYM_LANES=1 OPT_AMD_ARAP_VGRID=1536
} 2>&1 | tee gpurun_out/r03ad/config4_grid.txt
