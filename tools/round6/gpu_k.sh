#!/bin/bash
run() { OPT_AMD_CONFIG="$1" python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d.get('kernel_avg_us',{}); print(d['config'][:40], round(d['wall_s']*1e3,2),'ms', d['cost_final'], {n:round(k[n],2) for n in ('PCGStep1','PCGStep2+PCGStep3','PCGStep2','PCGStep3','packVertexRecords','vertexRecords') if n in k})
"; }
for v in "" _ellb1 _ellb2 _ellb4; do echo "=== variant '$v'"; OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt$v.so run config4; done
echo "=== volumetric"; run volumetric
