#!/bin/bash
# interleaved A/B of config 3 (SFS 1024^2 double LM 60x10) over environment settings:  tools/ab_sfs.sh "ENV=.. ENV=.." "..." ...
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for e in "$@"; do
  env $e OPT_AMD_CONFIG=config3 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_avg_us']; print('$e:', round(d['wall_s']*1e3,2), 'ms', d['outer_steps'], 'steps cost', d['cost_final'], {n:k[n] for n in k if n in ('PCGIteration','PCGStep1','PCGStep2','PCGStep3')})"
done
done
