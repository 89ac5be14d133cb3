"""BASELINE config 5 (image_warping 8192 x 8192 tiled across 8 GPUs) as far as a 1-GPU box allows ("missing" #2):

  * the 8192^2 image on ONE GPU: one Gauss-Newton step of 20 PCG iterations against the OpenMP oracle (cost at the float bar) -- the row-marching
    kernels at their largest single-GPU size (195 rows per workgroup, 805 MB per solver vector);
  * the bench's own 8-rank path at that size -- `bench.py --gpus 8 --share-gpu --size 8192`: launcher, 8 processes, 1024-row slabs with 8 ghost rows, every
    rank building only its own rows, peer-mailbox communicator over hipIpc windows -- with all ranks on device 0, against the single-GPU solve of the same
    problem.  Only the xGMI hop itself is not exercised.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import hip_solver

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE = 8192


def _single_gpu_cost(liters, steps=1):
    import torch
    P = wl.image_warping(SIZE, SIZE)
    g = hip_solver(P, nIterations=steps, lIterations=liters)
    dev = api.to_device(P)
    g.init(dev)
    costs = [g.cost()]
    while g.step(dev):
        costs.append(g.cost())
    torch.cuda.synchronize()
    g.close()
    del dev
    torch.cuda.empty_cache()
    return costs


def test_config5_8192_one_gpu_step_vs_oracle(oracle_lib):
    liters = 20
    P = wl.image_warping(SIZE, SIZE)
    o = oracle_lib.OracleSolver("image_warping", "gaussNewtonGPU", False, P.dims)
    o.set_threads(min(os.cpu_count() or 1, 128))
    o.set("nIterations", 1); o.set("lIterations", liters)
    o.init(P.params)
    ref = [o.cost()]
    o.step(P.params)
    ref.append(o.cost())
    o.close()
    del P
    costs = _single_gpu_cost(liters)
    assert abs(costs[0] - ref[0]) <= 1e-6 * ref[0]
    # 20 float PCG iterations on 67 M pixels: the bar of the 2048^2 / 4096^2 tests of this horizon
    assert abs(costs[1] - ref[1]) <= 3e-5 * ref[1], (costs, ref)


def test_config5_bench_8_ranks_sharing_the_gpu_equals_the_single_gpu_solve():
    liters = 14       # two halo-exchange periods of the 8-ghost-row slabs
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OPT_AMD_PEER_TIMEOUT="60")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--size", str(SIZE), "--steps", "1", "--warmup", "0",
           "--liters", str(liters), "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 8 and out["comm_ranks"] == 8, out
    assert "comm=peer" in out["config"]["parallelism"], out["config"]          # no silent fall-back to RCCL
    assert "8192" in out["config"]["workload"]
    single = _single_gpu_cost(liters)
    assert abs(out["cost_initial"] - single[0]) <= 1e-6 * single[0]
    assert np.isfinite(out["cost_final"]) and abs(out["cost_final"] - single[1]) <= 1e-5 * single[1], (out["cost_final"], single)
