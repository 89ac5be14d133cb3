#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests/test_energies_gpu.py tests/test_golden.py tests/test_steady_state_gpu.py tests/test_fullsize_gpu.py tests/test_cpp_callers_gpu.py tests/test_io.py -m gpu -q --timeout 600 -p no:cacheprovider -k "arap or config4 or raptor or golden" > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -10; grep -E "^E " $O/pytest.log | head -8
for i in 1 2; do
  OPT_AMD_CONFIG=config4 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_avg_us']; print('2-kernel', round(d['wall_s']*1e3,2), d['cost_final'], {x:k[x] for x in k if 'Step' in x})"
  OPT_AMD_ARAP_ITER=0 OPT_AMD_CONFIG=config4 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_avg_us']; print('3-kernel', round(d['wall_s']*1e3,2), d['cost_final'], {x:k[x] for x in k if 'Step' in x})"
done
