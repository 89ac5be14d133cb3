#!/bin/bash
# round 4, call g: communicator changes (sticky errors, shared-device detection, timing), self-describing N > 1 bench
mkdir -p gpurun_out/r04g
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_peer_comm_gpu.py tests/test_slab_gpu.py tests/test_config5_gpu.py tests/test_energies_gpu.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r04g/pytest.log
tail -25 gpurun_out/r04g/pytest.log
timeout 600 python bench.py --gpus 2 --share-gpu --size 2048 --steps 2 --warmup 1 --liters 50 --no-cpu-baseline > gpurun_out/r04g/bench_2ranks_shared.json 2> gpurun_out/r04g/bench_2ranks_shared.err
python - <<'PY'
import json
b = json.load(open("gpurun_out/r04g/bench_2ranks_shared.json"))
print(json.dumps({k: b[k] for k in ("value", "per_iteration_ms", "preflight", "rccl_leg")}, indent=1))
print(json.dumps(b["roofline"]["comm_kernels"], indent=1))
PY
tail -3 gpurun_out/r04g/bench_2ranks_shared.err
