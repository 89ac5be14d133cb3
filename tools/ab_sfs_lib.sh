#!/bin/bash
# tools/ab_sfs_lib.sh <lib1> <lib2> ...: config 3 over library variants (opt_amd/lib/<name>), interleaved
cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in "$@"; do
  OPT_AMD_LIB=$GRAFT_REPO_ROOT/opt_amd/lib/$lib OPT_AMD_CONFIG=config3 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_avg_us']; print('$lib:', round(d['wall_s']*1e3,2), 'ms', {n:k[n] for n in k if n in ('PCGIteration','PCGStep1')})"
done; done
