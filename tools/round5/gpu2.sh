#!/bin/bash
# round 5, GPU call 2: LM on-chip parity (full), flow-level oracle checks of the C++ callers, tests touched by the prune, HIP ensemble + L = 50 traces
mkdir -p gpurun_out/r05b
export PYTHONFAULTHANDLER=1
python -u -m pytest tests/test_onchip_lm_gpu.py tests/test_cpp_callers_gpu.py -x -q -m gpu > gpurun_out/r05b/lm_callers.txt 2>&1; echo "rc=$?" >> gpurun_out/r05b/lm_callers.txt
python -u -m pytest tests/test_lm_controls_gpu.py tests/test_image_warping_gpu.py tests/test_slab_gpu.py tests/test_energies_gpu.py -x -q -m gpu > gpurun_out/r05b/prune.txt 2>&1; echo "rc=$?" >> gpurun_out/r05b/prune.txt
python -u -m pytest tests/test_peer_comm_gpu.py -x -q -m gpu -k "posted" > gpurun_out/r05b/peer.txt 2>&1; echo "rc=$?" >> gpurun_out/r05b/peer.txt
python -u tools/horizon_ensemble.py --out gpurun_out/r05b/ensemble --traces 50 > gpurun_out/r05b/ensemble.log 2>&1; echo "rc=$?" >> gpurun_out/r05b/ensemble.log
for f in lm_callers prune peer; do echo "== $f"; tail -n 25 gpurun_out/r05b/$f.txt; done
tail -n 40 gpurun_out/r05b/ensemble.log
