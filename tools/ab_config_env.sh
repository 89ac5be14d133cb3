#!/bin/bash
# tools/ab_config_env.sh <config substring> "ENV=.." "ENV=.." ...: one config of tools/bench_configs.py over environment settings, interleaved
cfg=$1; shift
cd $GRAFT_REPO_ROOT
for r in 1 2; do for e in "$@"; do
  env $e OPT_AMD_CONFIG="$cfg" python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_avg_us']; print('$e:', round(d['wall_s']*1e3,2), 'ms cost', d['cost_final'], {n:k[n] for n in k if 'Step' in n})"
done; done
