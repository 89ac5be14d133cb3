"""GPU tests (-m gpu) of the row-slab tiling: the same problem solved as ONE image and as 2 / 3 / 4 slabs
(ranks = threads of this process sharing the GPU, in-process OptAmd_SlabComm) must give the same cost
trajectory and unknowns.  Exercises ghost rows, halo exchange of p / z / X, and the all-reduced sums."""
import numpy as np
import pytest

from opt_amd import api, slab, workloads as wl
from helpers import flat_unknowns, hip_solver, rel_err

pytestmark = pytest.mark.gpu


def _single(P, kind, **kw):
    g = hip_solver(P, kind, **kw)
    dev = api.to_device(P)
    g.init(dev); costs = [g.cost()]
    while g.step(dev):
        costs.append(g.cost())
    out = [dev[i].cpu().numpy() for i in P.unknown_slots]
    g.close()
    return costs, np.concatenate([o.reshape(-1) for o in out])


# ghost 1: PCG iteration with A*p in memory (iw_pcgIter); 2: without (iw_pcgIter2), r / p edge rows exchanged after every launch;
# 4, 8: the same kernel keeps ghost rows current by itself and the exchange happens every 3rd / 7th launch (14 launches: 4 / 2 exchanges)
@pytest.mark.parametrize("ghost", [1, 2, 4, 8])
@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("double", [False, True])
def test_gn_slabs_match_single(world, double, ghost):
    P = wl.image_warping(70, 53, double=double, random_state=3, mask_fraction=0.06, perturb=0.3)
    kw = dict(nIterations=3, lIterations=14)
    c1, x1 = _single(P.clone(), "gaussNewtonGPU", **kw)
    Q = P.clone()
    cN = slab.run_threads(Q, world, "gaussNewtonGPU", kw, ghost=ghost)
    tol = 1e-11 if double else 2e-5
    np.testing.assert_allclose(cN, c1, rtol=tol)
    assert rel_err(flat_unknowns(Q), x1) < tol


def test_lm_slabs_match_single():
    P = wl.image_warping(48, 40, double=True, random_state=5, mask_fraction=0.05, perturb=0.3)
    kw = dict(nIterations=4, lIterations=22)
    c1, x1 = _single(P.clone(), "LMGPU", **kw)
    Q = P.clone()
    cN = slab.run_threads(Q, 2, "LMGPU", kw)
    np.testing.assert_allclose(cN, c1, rtol=1e-9)
    assert rel_err(flat_unknowns(Q), x1) < 1e-8


def _rccl_world1(q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", OPT_AMD_FORCE_COMM="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    P = wl.image_warping(64, 40, random_state=3, perturb=0.3)
    job = slab.SlabJob("image_warping", 64, 40, 0, 1, problem=P.clone(), comm="rccl")
    assert job.comm_ranks() == 1
    job.solver.set_parameter("nIterations", 2); job.solver.set_parameter("lIterations", 10)
    job.solver.init(job.params); costs = [job.solver.cost()]
    while job.solver.step(job.params):
        costs.append(job.solver.cost())
    g = job.layout.ghost
    x = torch.cat([job.params[0][g:-g].reshape(-1), job.params[1][g:-g].reshape(-1)]).cpu().numpy()
    job.close()
    dist.destroy_process_group()
    q.put((costs, x))


def test_rccl_comm_single_rank():
    """The RCCL implementation of OptAmd_SlabComm (what bench.py --gpus N uses), driven with a 1-rank communicator:
    ncclCommInitRank from a broadcast id, in-stream ncclAllReduce of the PCG sums, grouped send/recv skipped at the
    image borders.  Must reproduce the plain single-GPU solve."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1, args=(q,))
    p.start()
    costs, x = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    P = wl.image_warping(64, 40, random_state=3, perturb=0.3)
    c1, x1 = _single(P, "gaussNewtonGPU", nIterations=2, lIterations=10)
    np.testing.assert_allclose(costs, c1, rtol=2e-5)
    assert rel_err(x, x1) < 2e-5


def test_deep_ghost_slabs_general_ur_shape_and_period_switch(monkeypatch):
    """5 ghost rows (period 4) with a jittered UrShape (the general kernel, which rebuilds M_a from the pairs it evaluates -- also on the ghost rows it updates), and the same slabs with OPT_AMD_SLAB_PERIOD=1 (exchange after every launch): identical results."""
    P = wl.image_warping(66, 47, double=True, random_state=13, mask_fraction=0.05, perturb=0.3, jitter_urshape=0.2)
    kw = dict(nIterations=2, lIterations=11)
    c1, x1 = _single(P.clone(), "gaussNewtonGPU", **kw)
    Q = P.clone()
    cN = slab.run_threads(Q, 3, "gaussNewtonGPU", kw, ghost=5)
    np.testing.assert_allclose(cN, c1, rtol=1e-11)
    assert rel_err(flat_unknowns(Q), x1) < 1e-11
    monkeypatch.setenv("OPT_AMD_SLAB_PERIOD", "1")
    R = P.clone()
    cR = slab.run_threads(R, 3, "gaussNewtonGPU", kw, ghost=5)
    np.testing.assert_allclose(cR, cN, rtol=1e-13)


def test_gn_slabs_lattice_and_general_ur_shape():
    """Ap-free iteration on slabs for both kernels: unit-lattice UrShape (M from the flag byte) and a
    jittered UrShape (M_O from the flag byte, M_a rebuilt from the pairs), odd row counts so the slabs differ in height."""
    for jitter in (0.0, 0.2):
        P = wl.image_warping(66, 47, random_state=11, mask_fraction=0.05, perturb=0.3, jitter_urshape=jitter)
        kw = dict(nIterations=2, lIterations=25)
        c1, x1 = _single(P.clone(), "gaussNewtonGPU", **kw)
        Q = P.clone()
        cN = slab.run_threads(Q, 3, "gaussNewtonGPU", kw)
        np.testing.assert_allclose(cN, c1, rtol=5e-5)
        assert rel_err(flat_unknowns(Q), x1) < 5e-5
