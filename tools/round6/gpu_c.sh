#!/bin/bash
# round 6, call C: the foreign-tenant test again (occupier now holds LDS), ARAP's Levenberg-Marquardt two-kernel iteration, the LM on-chip variants after the cos / sin move,
# and the SQ / TCC / TCP counter passes of config 4's two kernels
out=gpurun_out/r06c; mkdir -p $out
timeout 900 python -m pytest tests/test_coresidency_gpu.py tests/test_lm_controls_gpu.py tests/test_onchip_lm_gpu.py "tests/test_energies_gpu.py" -m gpu -q -x > $out/tests.txt 2>&1
tail -n 15 $out/tests.txt
OPT_AMD_CONFIG="arap" python tools/bench_configs.py > $out/arap_configs.json 2> $out/arap_configs.err; tail -c 1500 $out/arap_configs.json
bash tools/round6/arap_pmc.sh $out/arap_pmc config4 > $out/arap_pmc.log 2>&1
tail -n 80 $out/arap_pmc.log
