#!/bin/bash
# round-3 GPU batch B: horizon table after the exact-product expansion sums, peer tests with the posted all-reduce, slab overhead, bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03b; mkdir -p $O
python tools/horizon_parity.py --out $O/horizon > $O/horizon.log 2>&1; echo "horizon rc=$?"
timeout 900 python -m pytest tests/test_horizon_gpu.py tests/test_peer_comm_gpu.py tests/test_slab_gpu.py tests/test_steady_state_gpu.py tests/test_image_warping_gpu.py tests/test_config5_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -30
timeout 300 python tools/slab_overhead.py > $O/slab_overhead.txt 2>&1; echo "slab rc=$?"; cat $O/slab_overhead.txt | grep -v "^{" | tail -12
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03b/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["parity"]["rel_err"], d["gn_solve"])
PY
