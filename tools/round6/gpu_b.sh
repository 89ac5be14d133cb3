#!/bin/bash
# round 6, call B: the new tests first, then the whole GPU suite with the parity log on (input of tools/make_parity_bars.py), then the bench line
out=gpurun_out/r06b; mkdir -p $out
rm -f $out/parity_log.jsonl
timeout 600 python -m pytest tests/test_reference_order_mode_gpu.py tests/test_coresidency_gpu.py -m gpu -q -x > $out/new_tests.txt 2>&1
tail -n 25 $out/new_tests.txt
export OPT_PARITY_LOG=$PWD/$out/parity_log.jsonl
timeout 1500 python -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1
unset OPT_PARITY_LOG
tail -n 40 $out/gpu_suite.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.err
python - <<'PY'
import json
b=json.load(open("gpurun_out/r06b/bench.json"))
print({k:b[k] for k in ("value","ms_per_step")}); print("roofline", {k:b["roofline"].get(k) for k in ("frac","avg_kernel_ms","hbm_frac","frac_of_box_copy_float4")})
print("contract", json.dumps(b.get("contract_loop"))[:1800]); print("box", b.get("box"))
PY
