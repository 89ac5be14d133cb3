#!/bin/bash
out=gpurun_out/r06y; mkdir -p $out
timeout 1200 python -m pytest tests/test_onchip_stencil_gpu.py tests/test_onchip_sfs_gpu.py tests/test_onchip_lm_gpu.py tests/test_onchip_gpu.py tests/test_coresidency_gpu.py tests/test_lm_controls_gpu.py -m gpu -q -x 2>&1 | grep "passed\|failed\|^FAILED" | tail -3
bash tools/round6/artifacts.sh r06 > $out/artifacts.log 2>&1
tail -n 12 $out/artifacts.log | cut -c1-200
