#!/bin/bash
# round 4, call e: inbox collected inside the sum wait, flat batched sums, delta one chunk ahead
mkdir -p gpurun_out/r04e
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_onchip_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04e/pytest_onchip.log
tail -4 gpurun_out/r04e/pytest_onchip.log
OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt_prof.so OPT_AMD_ONCHIP_PROFILE=1 timeout 300 python tools/onchip_bench.py --liters 400 --steps 1 --sizes 512x512,1024x1024,4096x512 2>&1 | grep -A10 "on-chip profile" > gpurun_out/r04e/profile_all.txt
python - <<'PY'
# keep the last profile block of every size
blocks, cur = {}, None
for line in open("gpurun_out/r04e/profile_all.txt"):
    if line.startswith("on-chip profile"):
        cur = line.split()[2]; blocks[cur] = [line]
    elif cur and (line.startswith("   ")):
        blocks[cur].append(line)
open("gpurun_out/r04e/profile.txt", "w").write("".join("".join(b) for b in blocks.values()))
PY
cat gpurun_out/r04e/profile.txt
timeout 300 python tools/onchip_bench.py --liters 400 --steps 4 2>/dev/null > gpurun_out/r04e/onchip_sizes.md
cat gpurun_out/r04e/onchip_sizes.md
