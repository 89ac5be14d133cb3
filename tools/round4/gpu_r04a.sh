#!/bin/bash
# round 4, call a: first run of the on-chip linear solve -- parity tests, then the size table
mkdir -p gpurun_out/r04a
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_onchip_gpu.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r04a/pytest_onchip.log
tail -15 gpurun_out/r04a/pytest_onchip.log
timeout 300 python tools/onchip_bench.py --liters 400 --steps 4 > gpurun_out/r04a/onchip_sizes.md 2> gpurun_out/r04a/onchip_sizes.err
cat gpurun_out/r04a/onchip_sizes.md; tail -5 gpurun_out/r04a/onchip_sizes.err
