#!/bin/bash
# round 4, call b: where does an on-chip iteration's time go (OC_PROFILE build), and the size table with all of delta requested before the wait for the sums
mkdir -p gpurun_out/r04b
cd /root/repo
export PYTHONUNBUFFERED=1
OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt_prof.so OPT_AMD_ONCHIP_PROFILE=1 timeout 300 python tools/onchip_bench.py --liters 400 --steps 2 2>&1 | grep -v amdgpu.ids > gpurun_out/r04b/profile.txt
grep -E "on-chip profile" gpurun_out/r04b/profile.txt | sort | uniq -c | sort -rn | awk '{$1=""; print}' | sort -t' ' -k4 | awk '!seen[$4]++' 
timeout 300 python tools/onchip_bench.py --liters 400 --steps 4 2>/dev/null > gpurun_out/r04b/onchip_sizes.md
cat gpurun_out/r04b/onchip_sizes.md
timeout 600 python -m pytest tests/test_onchip_gpu.py -x -q -m gpu 2>&1 | tail -5
