#!/bin/bash
for v in "" _slp "" _slp; do echo "== variant '$v'"; OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt$v.so OPT_AMD_CONFIG="config4" python tools/bench_configs.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d.get('kernel_avg_us',{}); print(round(d['wall_s']*1e3,2),'ms', d['cost_final'], {n:round(k[n],2) for n in ('PCGStep1','PCGStep2+PCGStep3') if n in k})
"; done
