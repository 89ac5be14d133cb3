#!/bin/bash
# Interleaved A/B of library variants on one box:  tools/ab.sh <rounds> <lib1> <lib2> ...   (names under opt_amd/lib/)
rounds=$1; shift
cd $GRAFT_REPO_ROOT
for r in $(seq $rounds); do
  for lib in "$@"; do
    v=$(OPT_AMD_LIB=$GRAFT_REPO_ROOT/opt_amd/lib/$lib timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s  iter %.1f us' % (d['value'], d['roofline']['avg_kernel_ms']*1e3))")
    echo "$lib: $v"
  done
done
