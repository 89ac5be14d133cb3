#!/bin/bash
mkdir -p gpurun_out/r05t
timeout 600 python -m pytest tests/test_onchip_stencil_gpu.py -q -m gpu -k "intrinsic" --maxfail=8 -p no:cacheprovider > gpurun_out/r05t/pytest_in.txt 2>&1; echo "rc=$?" >> gpurun_out/r05t/pytest_in.txt
grep -v "^$" gpurun_out/r05t/pytest_in.txt | grep -v "^E   " | tail -n 30 | cut -c1-300
