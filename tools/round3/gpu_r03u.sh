#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests/test_energies_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "arap" > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -10; grep -E "^E " $O/pytest.log | head -8
