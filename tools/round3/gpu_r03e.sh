#!/bin/bash
# round-3 GPU batch E: SFS marching kernel (two-buffer loop) timing; in-kernel posted all-reduce: peer tests + slab overhead
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_lm_controls_gpu.py tests/test_energies_gpu.py tests/test_golden.py tests/test_steady_state_gpu.py tests/test_fullsize_gpu.py tests/test_cpp_callers_gpu.py tests/test_peer_comm_gpu.py tests/test_slab_gpu.py tests/test_config5_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "sfs or config3 or golden or peer or slab or posted or config5" > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -30
for i in 1 2; do
  OPT_AMD_CONFIG=config3 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('march', d['wall_s'], d['cost_final'], d['kernel_avg_us'].get('PCGIteration'))"
  OPT_AMD_SFS_MARCH=0 OPT_AMD_CONFIG=config3 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiled', d['wall_s'], d['cost_final'], d['kernel_avg_us'].get('PCGIteration'))"
done
for g in 256 512 768 1536 2048; do
  OPT_AMD_SFS_MARCH_GRID=$g OPT_AMD_CONFIG=config3 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('march grid $g', d['wall_s'], d['kernel_avg_us'].get('PCGIteration'))"
done
timeout 400 python tools/slab_overhead.py > $O/slab_overhead.txt 2>&1; echo "slab rc=$?"; grep "us per" $O/slab_overhead.txt
