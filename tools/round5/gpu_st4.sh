#!/bin/bash
mkdir -p gpurun_out/r05t
timeout 600 python -m pytest tests/test_onchip_stencil_gpu.py -q -m gpu -x -s -k "test_lm" -p no:cacheprovider > gpurun_out/r05t/pytest_lm.txt 2>&1; echo "rc=$?" >> gpurun_out/r05t/pytest_lm.txt
grep -v "^$" gpurun_out/r05t/pytest_lm.txt | tail -n 40 | cut -c1-400
