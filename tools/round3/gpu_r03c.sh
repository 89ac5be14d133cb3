#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c; mkdir -p $O
python tools/trace_compare.py --family adversarial --liters 40 > $O/trace_adv_float.txt 2>&1; echo "trace rc=$?"
python tools/trace_compare.py --family adversarial --liters 40 --double > $O/trace_adv_double.txt 2>&1
for i in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exact', d['value'])"
  OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt_floatsums.so python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('float', d['value'])"
done
head -60 $O/trace_adv_float.txt
