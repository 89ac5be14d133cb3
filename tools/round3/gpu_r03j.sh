#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03j; mkdir -p $O
timeout 900 python -m pytest tests/test_steady_state_gpu.py tests/test_image_warping_gpu.py tests/test_slab_gpu.py tests/test_horizon_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -30
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03j/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["general_urshape"], d["gn_solve"]["gn_solve_ms"], d["gn_solve"].get("rel_err_vs_oracle_float"), d["gn_solve"]["reference_default_10x10"])
PY
