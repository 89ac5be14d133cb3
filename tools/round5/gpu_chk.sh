#!/bin/bash
mkdir -p gpurun_out/r05c
timeout 900 python -m pytest tests/test_peer_comm_gpu.py tests/test_slab_gpu.py tests/test_energies_gpu.py -q -m gpu --maxfail=5 -p no:cacheprovider > gpurun_out/r05c/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r05c/pytest.txt
tail -n 5 gpurun_out/r05c/pytest.txt
