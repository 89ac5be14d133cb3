#!/bin/bash
mkdir -p gpurun_out/r05s
timeout 900 python -m pytest tests/test_onchip_sfs_gpu.py tests/test_onchip_lm_gpu.py tests/test_lm_controls_gpu.py -q -m gpu --maxfail=12 -p no:cacheprovider > gpurun_out/r05s/pytest2.txt 2>&1; echo "rc=$?" >> gpurun_out/r05s/pytest2.txt
tail -n 6 gpurun_out/r05s/pytest2.txt
bash tools/round5/gpu_sfs2.sh | grep wall
