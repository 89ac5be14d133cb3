#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03k; mkdir -p $O
timeout 900 python -m pytest tests/test_image_warping_gpu.py tests/test_steady_state_gpu.py tests/test_cpp_callers_gpu.py tests/test_slab_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -30
timeout 600 python bench.py --gpus 2 --share-gpu --size 2048 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_extras.json 2> $O/bench_2ranks_extras.err; echo "bench 2 ranks rc=$?"
tail -c 1500 $O/bench_2ranks_extras.json; tail -5 $O/bench_2ranks_extras.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03k/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["gn_solve"]["reference_default_10x10"], d["roofline"]["kernel_ms_per_step"])
PY
