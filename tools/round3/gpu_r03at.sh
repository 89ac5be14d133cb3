#!/bin/bash
# stencil functor engine: two-kernel Gauss-Newton iteration (OPT_AMD_SE_ITER) -- parity tests, then the three configs with and without
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03at
timeout 600 python -m pytest tests/test_energies_gpu.py tests/test_golden.py -m gpu -q -k "flow or intrinsic or volumetric" 2>&1 | grep -v "cost\|^ *$" | tail -8 > gpurun_out/r03at/pytest.log; cat gpurun_out/r03at/pytest.log
show='
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l); k = d["kernel_avg_us"]
        print("wall %.2f ms  cost %.9g  " % (d["wall_s"] * 1e3, d["cost_final"]), {n: round(k[n], 1) for n in k if "Step" in n})
'
run() { echo "== $1 $2"; env $2 OPT_AMD_CONFIG="$1" timeout 200 python tools/bench_configs.py 2>/dev/null | python -c "$show"; }
{
for c in optical_flow intrinsic volumetric; do for m in 1 0 1 0; do run $c OPT_AMD_SE_ITER=$m; done; done
} 2>&1 | tee gpurun_out/r03at/se_iter.txt
