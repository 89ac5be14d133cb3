#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03n; mkdir -p $O
timeout 900 python -m pytest tests/test_peer_comm_gpu.py tests/test_slab_gpu.py tests/test_config5_gpu.py tests/test_horizon_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|solve8" $O/pytest.log | sed 's/ - .*//' | tail -30
