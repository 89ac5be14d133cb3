#!/bin/bash
# the whole GPU suite + smoke (what the driver runs at round end)
mkdir -p gpurun_out/full
cd /root/repo
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/full/pytest_gpu.log
tail -15 gpurun_out/full/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
