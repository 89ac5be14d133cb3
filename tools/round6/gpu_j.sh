#!/bin/bash
run() { OPT_AMD_CONFIG="config4" python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d.get('kernel_avg_us',{}); print(round(d['wall_s']*1e3,2),'ms', d['cost_final'], {n:round(k[n],2) for n in ('PCGStep1','PCGStep2+PCGStep3','packVertexRecords','vertexRecords') if n in k})
"; }
for w in 1 0; do for x in 1 0; do for g in 0 512 768 1024; do
  echo "=== walk $w xcd $x vgrid $g"
  if [ $g = 0 ]; then OPT_AMD_ARAP_WALK=$w OPT_AMD_ARAP_SYM_XCD=$x run; else OPT_AMD_ARAP_WALK=$w OPT_AMD_ARAP_SYM_XCD=$x OPT_AMD_ARAP_VGRID=$g run; fi
done; done; done
timeout 600 python -m pytest tests/test_energies_gpu.py tests/test_lm_controls_gpu.py tests/test_config_horizon_gpu.py -m gpu -q -k "arap or volumetric or config4" 2>&1 | tail -3
