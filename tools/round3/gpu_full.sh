#!/bin/bash
# the driver's round-end checks on one box: pytest -m gpu, smoke(), bench with the driver's arguments
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/full; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | sed 's/ - .*//' | tail -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads([l for l in open('gpurun_out/full/bench.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['hbm_frac'], d['gn_solve_ms'], d['general_urshape']['value'], d['cpu_baseline']['value'])"
