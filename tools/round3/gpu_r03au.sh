#!/bin/bash
# volumetric_mesh_deformation on ARAP's kernel set (lattice graph) against the functor engine: parity tests both ways, then the 96^3 config
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03au
for m in 1 0; do OPT_AMD_VOLUMETRIC_ARAP=$m timeout 600 python -m pytest tests/test_energies_gpu.py tests/test_golden.py -m gpu -q -k "volumetric" 2>&1 | grep -v "cost\|^ *$" | tail -4; done | tee gpurun_out/r03au/pytest.log
show='
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l); k = d["kernel_avg_us"]
        print("wall %.2f ms  cost %.9g  " % (d["wall_s"] * 1e3, d["cost_final"]), {n: round(k[n], 1) for n in k if "Step" in n})
'
run() { echo "== $1 $2"; env $2 OPT_AMD_CONFIG="$1" timeout 200 python tools/bench_configs.py 2>/dev/null | python -c "$show"; }
{ for m in 1 0 1 0; do run volumetric OPT_AMD_VOLUMETRIC_ARAP=$m; done; } 2>&1 | tee gpurun_out/r03au/volumetric.txt
