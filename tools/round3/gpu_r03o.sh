#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03o; mkdir -p $O
timeout 900 python -m pytest tests/test_image_warping_gpu.py tests/test_steady_state_gpu.py tests/test_slab_gpu.py tests/test_peer_comm_gpu.py tests/test_horizon_gpu.py tests/test_lm_controls_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -30
for sz in 2048 4096; do
for i in 1 2; do
  python bench.py --size $sz --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sz shared', d['value'], d['cost_final'])"
  OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt_nopairshare.so python bench.py --size $sz --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sz unshared', d['value'], d['cost_final'])"
done; done
timeout 300 python tools/slab_overhead.py 2>/dev/null | grep "us per" 
