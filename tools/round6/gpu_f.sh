#!/bin/bash
# round 6, call F: ARAP with the {p, M, U} records -- parity tests, config 4 per variant, the LM row, and the co-residency tests
out=gpurun_out/r06f; mkdir -p $out
timeout 900 python -m pytest tests/test_energies_gpu.py tests/test_lm_controls_gpu.py tests/test_reference_order_mode_gpu.py tests/test_coresidency_gpu.py tests/test_fullsize_gpu.py "tests/test_steady_state_gpu.py::test_config4_arap_500k_step_vs_oracle" -m gpu -q -k "arap or volumetric or coresidency or config4 or reference_order" > $out/tests.txt 2>&1
grep -n "passed\|failed\|^FAILED" $out/tests.txt | tail
for v in "" _arapb2 _arapb3 _arapl1b3 _arapl1b6 _arapl4b1; do
  echo "=== variant '$v'"
  OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt$v.so OPT_AMD_CONFIG="config4" python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d.get('kernel_avg_us',{}); print(round(d['wall_s']*1e3,2),'ms', d['cost_final'], {n:round(k[n],2) for n in ('PCGStep1','PCGStep2+PCGStep3','packVertexRecords','vertexRecords') if n in k})
"
done
