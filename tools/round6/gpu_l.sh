#!/bin/bash
# round 6, call L: the whole GPU suite with the parity log on (bars in force from the first log), then smoke and the bench line
out=gpurun_out/r06l; mkdir -p $out
rm -f $out/parity_log.jsonl
export OPT_PARITY_LOG=$PWD/$out/parity_log.jsonl
timeout 1800 python -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1
unset OPT_PARITY_LOG
grep -n "passed\|failed\|^FAILED\|^ERROR" $out/gpu_suite.txt | tail -20
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -n 2 $out/smoke.txt
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
b=json.load(open("gpurun_out/r06l/bench.json"))
print({k:b[k] for k in ("value","ms_per_step")}); print("roofline", {k:b["roofline"].get(k) for k in ("frac","avg_kernel_ms","hbm_frac","frac_of_box_copy_float4")})
c=b.get("contract_loop") or {}
print("contract", {k:c.get(k) for k in ("pcg_iters_per_s","ms_per_step","within_contract_1e-5_of_fma_oracle")}, c.get("roofline",{}).get("frac"), c.get("roofline",{}).get("frac_physical"), (c.get("rel_err_vs_fma_oracle") or {}).get("after_1_step_400_pcg"), (c.get("rel_err_vs_fma_oracle") or {}).get("after_8_steps"))
print("box", b.get("box",{}).get("kind"))
PY
