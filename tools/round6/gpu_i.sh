#!/bin/bash
out=gpurun_out/r06i; mkdir -p $out
timeout 900 python -m pytest tests/test_energies_gpu.py tests/test_lm_controls_gpu.py tests/test_reference_order_mode_gpu.py tests/test_fullsize_gpu.py tests/test_config_horizon_gpu.py "tests/test_steady_state_gpu.py::test_config4_arap_500k_step_vs_oracle" tests/test_cpp_callers_gpu.py -m gpu -q -k "arap or volumetric or config4 or reference_order or config3" > $out/tests.txt 2>&1
grep -n "passed\|failed\|^FAILED" $out/tests.txt | tail
run() { OPT_AMD_CONFIG="config4" python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d.get('kernel_avg_us',{}); print(round(d['wall_s']*1e3,2),'ms', d['cost_final'], {n:round(k[n],2) for n in ('PCGStep1','PCGStep2+PCGStep3','packVertexRecords','vertexRecords') if n in k})
"; }
for v in "" _ellb1 _ellb2 _ellb6; do
  for g in 0 768 1024 1536; do
    echo "=== variant '$v' vgrid $g"
    if [ $g = 0 ]; then OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt$v.so run; else OPT_AMD_ARAP_VGRID=$g OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt$v.so run; fi
  done
done
