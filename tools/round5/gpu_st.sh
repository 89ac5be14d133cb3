#!/bin/bash
mkdir -p gpurun_out/r05t
timeout 1200 python -m pytest tests/test_onchip_stencil_gpu.py -q -m gpu --maxfail=15 -p no:cacheprovider > gpurun_out/r05t/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r05t/pytest.txt
grep -v "^E  \|^$" gpurun_out/r05t/pytest.txt | tail -n 40
