#!/bin/bash
# image_warping iteration kernel: cos / sin from the 8 B/px table (CSTAB) against the 4 B/px angle + inline sincos, by image size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03an
timeout 600 python -m pytest tests/test_image_warping_gpu.py tests/test_steady_state_gpu.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r03an/pytest.log; cat gpurun_out/r03an/pytest.log
timeout 600 bash tools/ab_env_sizes.sh OPT_AMD_CSTAB=0 OPT_AMD_CSTAB=1 2>&1 | tee gpurun_out/r03an/cstab_sizes.txt
