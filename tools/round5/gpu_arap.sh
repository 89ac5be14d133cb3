#!/bin/bash
# round 5: the ARAP on-chip experiment (development variant libOpt_arapexp.so) -- phase profile + config 4 timing
mkdir -p gpurun_out/r05f
[ -f opt_amd/lib/libOpt_arapexp.so ] || python -c 'from opt_amd import build; build.build_variant("arapexp", ["OPT_AMD_ARAP_ONCHIP"])' > /dev/null      # (the development variant is not kept in the tree)
export OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt_arapexp.so
timeout 300 python -u tools/round5/dbg_arap.py 60 50 8 2 > gpurun_out/r05f/dbg.txt 2>&1; echo "rc=$?" >> gpurun_out/r05f/dbg.txt
OPT_AMD_ONCHIP_PROFILE=1 OPT_AMD_CONFIG=config4 timeout 600 python tools/bench_configs.py > gpurun_out/r05f/config4.json 2> gpurun_out/r05f/config4.err
tail -n 6 gpurun_out/r05f/dbg.txt; grep "profile" gpurun_out/r05f/config4.err | tail -n 3
python - <<'PY'
import json
for ln in open('gpurun_out/r05f/config4.json'):
    r=json.loads(ln); print(r["wall_s"], {k:v for k,v in r["kernel_avg_us"].items() if "PCG" in k or "Solve" in k or "build" in k})
PY
