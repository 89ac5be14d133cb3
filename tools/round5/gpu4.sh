#!/bin/bash
# round 5, GPU call 4: the rewritten long-horizon tests, LM variants, flow-level checks, dry runs, a short bench (box object, coarse events)
mkdir -p gpurun_out/r05d
export PYTHONFAULTHANDLER=1
timeout 1200 python -u -m pytest tests/test_horizon_gpu.py -x -q -m gpu -s > gpurun_out/r05d/horizon.txt 2>&1; echo "rc=$?" >> gpurun_out/r05d/horizon.txt
timeout 900 python -u -m pytest tests/test_onchip_lm_gpu.py tests/test_cpp_callers_gpu.py -x -q -m gpu > gpurun_out/r05d/lm_callers.txt 2>&1; echo "rc=$?" >> gpurun_out/r05d/lm_callers.txt
python bench.py --dry > gpurun_out/r05d/dry_1.json 2> gpurun_out/r05d/dry_1.err
timeout 300 python bench.py --gpus 2 --share-gpu --dry > gpurun_out/r05d/dry_2.json 2> gpurun_out/r05d/dry_2.err
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r05d/bench.json 2> gpurun_out/r05d/bench.err
grep -v "^E  \|^$" gpurun_out/r05d/horizon.txt | tail -n 40
tail -n 12 gpurun_out/r05d/lm_callers.txt
cut -c1-900 gpurun_out/r05d/dry_1.json; tail -c 1500 gpurun_out/r05d/dry_2.json; tail -3 gpurun_out/r05d/dry_2.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r05d/bench.json').read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step","box")})
r=b["roofline"]; print({k:r[k] for k in ("frac","avg_kernel_ms","kernel_ms_per_step_sum","timed_leg_ms_per_step","traffic")})
print(b["reference_example_flows"])
print(b["parity"]["rel_err"], b["parity"]["within_reference_spread"], b["gn_solve"].get("rel_err_vs_oracle_float"), b["gn_solve"].get("within_reference_spread"))
PY
