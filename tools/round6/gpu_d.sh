#!/bin/bash
out=gpurun_out/r06d; mkdir -p $out
timeout 900 python -m pytest tests/test_coresidency_gpu.py tests/test_lm_controls_gpu.py tests/test_onchip_lm_gpu.py tests/test_energies_gpu.py tests/test_stencil_march_gpu.py tests/test_onchip_stencil_gpu.py -m gpu -q > $out/tests.txt 2>&1
tail -n 15 $out/tests.txt
OPT_AMD_CONFIG="arap" python tools/bench_configs.py > $out/arap_configs.json 2> $out/arap_configs.err; tail -c 1800 $out/arap_configs.json
bash tools/round6/arap_pmc.sh $out/arap_pmc config4 > $out/arap_pmc.log 2>&1
du -sh $out
