#!/bin/bash
# Time one BASELINE config (tools/bench_configs.py substring filter) under several env settings / libs: tools/ab_config.sh <rounds> <config> "ENV=..:lib.so" ...
rounds=$1; cfg=$2; shift; shift
cd $GRAFT_REPO_ROOT
for r in $(seq $rounds); do
  for c in "$@"; do
    e=${c%%:*}; lib=${c##*:}
    v=$(env $e OPT_AMD_CONFIG="$cfg" OPT_AMD_LIB=$GRAFT_REPO_ROOT/opt_amd/lib/$lib timeout 200 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms  %.0f it/s  %s' % (d['wall_s']*1e3, d['pcg_iters_per_s_nominal'], {k:v for k,v in d['kernel_avg_us'].items() if 'Step' in k}))")
    echo "[$c]: $v"
  done
done
