#!/bin/bash
# round 4, call c: the on-chip kernel with ONE grid-wide wait per iteration -- parity tests, phase profile, size table, flat against tree sums
mkdir -p gpurun_out/r04c
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_onchip_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04c/pytest_onchip.log
tail -8 gpurun_out/r04c/pytest_onchip.log
OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt_prof.so OPT_AMD_ONCHIP_PROFILE=1 timeout 300 python tools/onchip_bench.py --liters 400 --steps 2 2>&1 | grep "on-chip profile" | awk '!seen[$4]++' > gpurun_out/r04c/profile.txt
cat gpurun_out/r04c/profile.txt
timeout 300 python tools/onchip_bench.py --liters 400 --steps 4 2>/dev/null > gpurun_out/r04c/onchip_sizes.md
cat gpurun_out/r04c/onchip_sizes.md
echo "flat sums up to 256 workgroups:"
OPT_AMD_ONCHIP_FLAT=256 timeout 300 python tools/onchip_bench.py --liters 400 --steps 4 2>/dev/null > gpurun_out/r04c/onchip_sizes_flat256.md
cat gpurun_out/r04c/onchip_sizes_flat256.md
