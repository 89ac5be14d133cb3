#!/bin/bash
# round 4, call f: delta two chunks ahead; bench.py with the on-chip table and the reference's example flows; C++ callers
mkdir -p gpurun_out/r04f
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_onchip_gpu.py tests/test_cpp_callers_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python tools/onchip_bench.py --liters 400 --steps 4 2>/dev/null > gpurun_out/r04f/onchip_sizes.md
cat gpurun_out/r04f/onchip_sizes.md
timeout 900 python bench.py --steps 6 --warmup 2 > gpurun_out/r04f/bench.json 2> gpurun_out/r04f/bench.err
python - <<'PY'
import json
b = json.load(open("gpurun_out/r04f/bench.json"))
print("value", b["value"], "gn_solve_ms", b["gn_solve_ms"])
print(json.dumps(b["onchip"], indent=1))
print(json.dumps(b["reference_example_flows"], indent=1))
PY
tail -3 gpurun_out/r04f/bench.err
