#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03p; mkdir -p $O
timeout 900 python -m pytest tests/test_lm_controls_gpu.py tests/test_energies_gpu.py tests/test_golden.py tests/test_steady_state_gpu.py tests/test_fullsize_gpu.py tests/test_cpp_callers_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "sfs or config3 or golden" > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -10
for i in 1 2 3; do
  OPT_AMD_CONFIG=config3 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3', d['wall_s'], d['kernel_avg_us'])"
done
timeout 600 python bench.py --gpus 8 --share-gpu --size 4096 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_8ranks_shared_gpu_4096.json 2> $O/b8a.err; echo "8 ranks 4096 rc=$?"
timeout 900 python bench.py --gpus 8 --share-gpu --size 8192 --steps 1 --warmup 1 --liters 100 --no-cpu-baseline --no-extras > $O/bench_8ranks_shared_gpu_8192.json 2> $O/b8b.err; echo "8 ranks 8192 rc=$?"
python -c "
import json
for f in ('bench_8ranks_shared_gpu_4096.json','bench_8ranks_shared_gpu_8192.json'):
    d=json.loads([l for l in open('gpurun_out/r03p/'+f) if l.startswith('{')][-1]); print(f, d['n_gpus'], d['comm_ranks'], d['config']['parallelism'], d['cost_final'], d['kernel_src_sha16'], d.get('gn_solve'))
"
