#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for sz in 4096 2048; do
for i in 1 2; do
  python bench.py --size $sz --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sz real', d['value'])"
  OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt_fakesincos.so python bench.py --size $sz --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sz fake', d['value'])"
done; done
