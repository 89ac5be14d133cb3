#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_lm_controls_gpu.py tests/test_energies_gpu.py tests/test_golden.py tests/test_steady_state_gpu.py tests/test_fullsize_gpu.py tests/test_cpp_callers_gpu.py tests/test_horizon_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "sfs or config3 or golden or horizon or solve" > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -30
for i in 1 2; do
  OPT_AMD_CONFIG=config3 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('march', d['wall_s'], d['cost_final'], d['kernel_avg_us'])"
  OPT_AMD_SFS_MARCH=0 OPT_AMD_CONFIG=config3 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiled', d['wall_s'], d['cost_final'], d['kernel_avg_us'].get('PCGIteration'))"
done
for g in 384 512 640 1024; do
  OPT_AMD_SFS_MARCH_GRID=$g OPT_AMD_CONFIG=config3 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('march grid $g', d['wall_s'], d['kernel_avg_us'].get('PCGIteration'))"
done
