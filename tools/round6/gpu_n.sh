#!/bin/bash
for x in 1 0 1 0; do
  echo "=== OPT_AMD_ITER_XCD=$x"
  OPT_AMD_ITER_XCD=$x python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['value'], b['ms_per_step'], b['parity']['rel_err'] if b.get('parity') else None)"
done
for s in 2048 8192; do for x in 1 0; do echo "=== size $s xcd $x"; OPT_AMD_ITER_XCD=$x python bench.py --size $s --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['value'], b['ms_per_step'])"; done; done
timeout 900 python -m pytest tests/test_steady_state_gpu.py tests/test_image_warping_gpu.py tests/test_slab_gpu.py -m gpu -q -x 2>&1 | grep "passed\|failed" | tail -3
