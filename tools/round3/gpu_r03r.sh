#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03r; mkdir -p $O
timeout 600 python -m pytest tests/test_image_warping_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "switches" > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -10; grep -E "^E " $O/pytest.log | head -12
