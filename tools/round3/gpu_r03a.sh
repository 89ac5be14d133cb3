#!/bin/bash
# round-3 GPU batch A: horizon parity table, full GPU test suite, bench (new and old once-per-step kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03a; mkdir -p $O
python tools/horizon_parity.py --out $O/horizon > $O/horizon.log 2>&1
echo "horizon rc=$?"
timeout 1100 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
OPT_AMD_MARCH_INIT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_oldinit.json 2> $O/bench_oldinit.err; echo "bench old rc=$?"
python - <<'PY'
import json
for f in ("bench.json", "bench_oldinit.json"):
    try:
        d = json.loads([l for l in open("gpurun_out/r03a/" + f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d.get("gn_solve"), (d.get("roofline") or {}).get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "unparsable", e)
PY
