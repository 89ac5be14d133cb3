#!/bin/bash
# round 5, GPU call 3: the persistent ARAP solve (parity + config 4 timing), float LM variants again
mkdir -p gpurun_out/r05c
export PYTHONFAULTHANDLER=1
timeout 600 python -u tools/round5/dbg_arap.py 41 29 > gpurun_out/r05c/dbg_arap.txt 2>&1; echo "rc=$?" >> gpurun_out/r05c/dbg_arap.txt
timeout 900 python -u -m pytest tests/test_arap_onchip_gpu.py -x -q -m gpu > gpurun_out/r05c/arap.txt 2>&1; echo "rc=$?" >> gpurun_out/r05c/arap.txt
timeout 900 python -u -m pytest tests/test_onchip_lm_gpu.py -x -q -m gpu > gpurun_out/r05c/lm.txt 2>&1; echo "rc=$?" >> gpurun_out/r05c/lm.txt
OPT_AMD_CONFIG=config4 timeout 600 python tools/bench_configs.py > gpurun_out/r05c/config4_onchip.json 2> gpurun_out/r05c/config4_onchip.err
OPT_AMD_ONCHIP=0 OPT_AMD_CONFIG=config4 timeout 600 python tools/bench_configs.py > gpurun_out/r05c/config4_stream.json 2> gpurun_out/r05c/config4_stream.err
for f in dbg_arap arap lm; do echo "== $f"; tail -n 25 gpurun_out/r05c/$f.txt; done
cat gpurun_out/r05c/config4_onchip.json gpurun_out/r05c/config4_stream.json | cut -c1-600
