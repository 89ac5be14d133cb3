#!/bin/bash
# A/B of library variants at a given image size: tools/ab_size.sh <rounds> <size> <lib1> <lib2> ...
rounds=$1; size=$2; shift; shift
cd $GRAFT_REPO_ROOT
for r in $(seq $rounds); do
  for lib in "$@"; do
    v=$(OPT_AMD_LIB=$GRAFT_REPO_ROOT/opt_amd/lib/$lib timeout 120 python bench.py --size $size --steps 2 --warmup 1 --liters 200 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s  iter %.1f us' % (d['value'], d['roofline']['avg_kernel_ms']*1e3))")
    echo "$lib @ $size: $v"
  done
done
