"""GPU tests (-m gpu) of the peer-mailbox communicator (opt_amd/csrc/comm/peer_comm.hip) with REAL processes: every rank is its own
process with its own HIP context, windows are exchanged as hipIpc handles and written by peer stores -- the code path bench.py --gpus N
takes on an 8-GPU node.  On a 1-GPU box the ranks share device 0 (two processes, one GPU): the IPC mapping, the sequence-numbered
mailbox all-reduce, the double-buffered halo staging with acknowledgements and the rank-ordered sums are all exercised; only the xGMI
hop itself is not.  Each rank's result must equal the single-GPU solve."""
import os

import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import hip_solver, rel_err

pytestmark = pytest.mark.gpu


def _single(P, kind, **kw):
    g = hip_solver(P, kind, **kw)
    dev = api.to_device(P)
    g.init(dev); costs = [g.cost()]
    while g.step(dev):
        costs.append(g.cost())
    out = [dev[i].cpu().numpy() for i in P.unknown_slots]
    g.close()
    return costs, out


def _rank_main(rank, world, port, case, q):
    import torch
    import torch.distributed as dist
    from opt_amd import slab
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OPT_AMD_PEER_TIMEOUT="8")
    # ranks SHARING one GPU: the iteration kernel polls the posted all-reduce in its prologue, so all ranks' kernels must be co-resident (on one GPU per rank they
    # trivially are): cap every rank's grid so that together they fit the chip's 256 CUs
    if not case.get("uncapped"):
        os.environ["OPT_AMD_ITER_MAXWG"] = str(max(1, 224 // world))
        os.environ["OPT_AMD_PEER_POST"] = "1"      # grids capped: keep the posted all-reduce although the ranks share a GPU (the communicator would take it away)
    os.environ.update(case.get("env", {}))
    os.environ.update(case.get("env_by_rank", {}).get(rank, {}))
    if world == 1:
        os.environ["OPT_AMD_FORCE_COMM"] = "1"
    torch.cuda.set_device(0)                                   # all ranks share the box's one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = wl.image_warping(case["W"], case["H"], double=case["double"], random_state=3, mask_fraction=0.06, perturb=0.3)
    job = slab.SlabJob("image_warping", case["W"], case["H"], rank, world, kind=case["kind"], double=case["double"], problem=P, ghost=case["ghost"], comm="peer")
    job.solver.set_parameter("nIterations", case["n"]); job.solver.set_parameter("lIterations", case["l"])
    if "expect_onchip" in case:
        job.solver.set_timing(True)
    job.solver.init(job.params); costs = [job.solver.cost()]
    while job.solver.step(job.params):
        costs.append(job.solver.cost())
    torch.cuda.synchronize()
    assert job.comm_kind == "peer" and job._peer.self_test_ok      # the self-test passed: no silent fall-back to RCCL in this test
    if "expect_onchip" in case:
        assert ("PCGSolveOnChip" in job.solver.kernel_timings()) == case["expect_onchip"], job.solver.kernel_timings().keys()
    if case.get("expect_fallback"):
        t = job.solver.kernel_timings()
        assert t["PCGSolveOnChip"][0] == 1 and "PCGIteration" in t and job.solver.on_chip_status() == 2, (t.keys(), job.solver.on_chip_status())
    q.put((rank, costs, job.owned_unknowns(), job.layout.row0, job.layout.rows, job._peer.mem_kind, job._peer.error()))
    job.close()
    dist.destroy_process_group()


def _run(world, case):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=300)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("ghost,double", [(8, False), (2, True), (1, True)])
def test_peer_mailbox_ranks_as_processes(world, ghost, double):
    """ghost 8: the deep-ghost protocol of bench.py (one r / p exchange per 7 launches); ghost 2: an exchange after every launch (the staging
    double buffer and its acknowledgements turn over 14 times per step); ghost 1: the kernel with A p in memory (A p rows exchanged)."""
    case = dict(W=70, H=64, double=double, ghost=ghost, kind="gaussNewtonGPU", n=3, l=14)
    P = wl.image_warping(case["W"], case["H"], double=double, random_state=3, mask_fraction=0.06, perturb=0.3)
    c1, x1 = _single(P, case["kind"], nIterations=case["n"], lIterations=case["l"])
    res = _run(world, case)
    tol = 1e-11 if double else 2e-5
    for r in range(world):
        _, costs, unk, row0, rows, mem_kind, err = res[r]
        assert err == 0 and mem_kind in ("uncached", "fine-grained")
        np.testing.assert_allclose(costs, c1, rtol=tol)                         # every rank sees the same global cost trajectory
        for a, b in zip(unk, x1):
            assert rel_err(a, b[row0:row0 + rows]) < tol
    if world == 2:
        assert res[0][1] == res[1][1]                                           # rank-ordered sums: bitwise identical on every rank


def test_ranks_sharing_a_gpu_run_uncapped_without_deadlock():
    """the posted all-reduce is polled by every workgroup of the next iteration kernel, which needs all ranks' kernels co-resident; ranks that share a
    GPU with full-chip grids are not.  The communicator sees its peers' PCI bus ids next to their IPC handles and, finding one on its own GPU, keeps the waiting
    all-reduce (one workgroup): two ranks with NO grid cap and NO switch set run to the single-GPU result."""
    case = dict(W=300, H=128, double=False, ghost=8, kind="gaussNewtonGPU", n=2, l=14, uncapped=True)
    P = wl.image_warping(case["W"], case["H"], double=False, random_state=3, mask_fraction=0.06, perturb=0.3)
    c1, x1 = _single(P, case["kind"], nIterations=case["n"], lIterations=case["l"])
    res = _run(2, case)
    for r in range(2):
        _, costs, unk, row0, rows, mem_kind, err = res[r]
        assert err == 0
        np.testing.assert_allclose(costs, c1, rtol=2e-5)
    assert res[0][1] == res[1][1]


@pytest.mark.parametrize("world,ghost,double", [(3, 2, True), (4, 8, False), (4, 1, True), (8, 8, False)])
def test_peer_mailbox_middle_ranks(world, ghost, double):
    """3 and 4 processes: the middle ranks have a neighbour on BOTH sides (two staging blocks pushed and pulled per exchange, acknowledgements from two
    peers) and the mailbox carries 3 - 4 contributions per all-reduce -- what every rank but the first and last of an 8-GPU run does."""
    case = dict(W=70, H=64 if world < 8 else 136, double=double, ghost=ghost, kind="gaussNewtonGPU", n=2, l=14)      # (8, 8): bench.py's rank count and ghost depth
    P = wl.image_warping(case["W"], case["H"], double=double, random_state=3, mask_fraction=0.06, perturb=0.3)
    c1, x1 = _single(P, case["kind"], nIterations=case["n"], lIterations=case["l"])
    res = _run(world, case)
    tol = 1e-11 if double else 2e-5
    for r in range(world):
        _, costs, unk, row0, rows, mem_kind, err = res[r]
        assert err == 0
        np.testing.assert_allclose(costs, c1, rtol=tol)
        assert costs == res[0][1]                                               # bitwise identical on every rank
        for a, b in zip(unk, x1):
            assert rel_err(a, b[row0:row0 + rows]) < tol


@pytest.mark.parametrize("world", [2, 4])
def test_posted_allreduce_equals_the_waited_one(world):
    """Round 3: the per-iteration all-reduce is POSTED by a one-workgroup kernel of the communicator's (k_mailPost, OptAmd_SlabCommExt.allReducePost) and the next
    iteration kernel's prologue polls this rank's mailbox (common.h pollMailSums) instead of a kernel waiting between two launches.  Same contributions, same order
    of the additions: the cost trajectory must be bitwise the one of the waited all-reduce, on every rank."""
    out = {}
    modes = {"posted": {"OPT_AMD_PEER_POST": "1"}, "waited": {"OPT_AMD_PEER_POST": "0"}}
    for name, env in modes.items():
        case = dict(W=70, H=96, double=False, ghost=8, kind="gaussNewtonGPU", n=2, l=30, env=env)
        res = _run(world, case)
        assert all(res[r][6] == 0 for r in range(world))
        assert all(res[r][1] == res[0][1] for r in range(world))
        out[name] = res[0][1]
    assert out["posted"] == out["waited"], out


def test_peer_mailbox_lm_two_processes():
    case = dict(W=48, H=40, double=True, ghost=2, kind="LMGPU", n=3, l=12)
    P = wl.image_warping(case["W"], case["H"], double=True, random_state=3, mask_fraction=0.06, perturb=0.3)
    c1, x1 = _single(P, "LMGPU", nIterations=3, lIterations=12)
    res = _run(2, case)
    for r in range(2):
        np.testing.assert_allclose(res[r][1], c1, rtol=1e-9)
        for a, b in zip(res[r][2], x1):
            assert rel_err(a, b[res[r][3]:res[r][3] + res[r][4]]) < 1e-8


def _fallback_main(q):
    import torch
    import torch.distributed as dist
    from opt_amd import slab
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OPT_AMD_FORCE_COMM="1", OPT_AMD_PEER_MEM="99")      # no such memory kind: window allocation fails
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    P = wl.image_warping(64, 40, random_state=3, perturb=0.3)
    job = slab.SlabJob("image_warping", 64, 40, 0, 1, problem=P.clone(), comm="peer")
    job.solver.set_parameter("nIterations", 2); job.solver.set_parameter("lIterations", 10)
    job.solver.init(job.params); costs = [job.solver.cost()]
    while job.solver.step(job.params):
        costs.append(job.solver.cost())
    q.put((job.comm_kind, costs))
    job.close()
    dist.destroy_process_group()


def test_peer_communicator_falls_back_to_rccl_when_unavailable():
    """A machine on which the window cannot be allocated / exported / mapped, or whose self-test (one all-reduce and one halo exchange with known
    answers) fails, runs the same job over RCCL instead of aborting -- the verdict is collective, so no rank is left waiting."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_fallback_main, args=(q,))
    p.start()
    kind, costs = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0 and kind == "rccl"
    P = wl.image_warping(64, 40, random_state=3, perturb=0.3)
    c1, _ = _single(P, "gaussNewtonGPU", nIterations=2, lIterations=10)
    np.testing.assert_allclose(costs, c1, rtol=2e-5)


def test_bench_two_ranks_is_self_describing():
    """the first N > 1 run must explain itself.  `bench.py --gpus 2 --share-gpu` (functional on a 1-GPU box): the JSON line carries a
    pre-flight block per rank (reachable devices, IPC window / open / self-test with its time, communicator chosen and why), rank 0's communicator kernel times
    per PCG iteration next to the per-iteration wall time, and the RCCL leg -- here the marker that ranks sharing a GPU cannot form an RCCL communicator."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OPT_AMD_PEER_TIMEOUT="60")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--size", "1024", "--steps", "1", "--warmup", "0", "--liters", "14", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["comm_ranks"] == 2
    pf = out["preflight"]
    assert len(pf) == 2 and [p["rank"] for p in pf] == [0, 1]
    for p in pf:
        assert p["comm"] == "peer" and p["why"] == "requested" and p["can_access_peer"][p["device"]] is True
        assert p["peer"]["ipc_window"] and p["peer"]["ipc_open"] and p["peer"]["self_test"] and p["peer"]["self_test_ms"] > 0
        assert p["peer"]["shares_device"] is True and p["peer"]["posted_allreduce"] is True      # bench.py --share-gpu caps the grids and says so (OPT_AMD_PEER_POST=1)
    ck = out["roofline"]["comm_kernels"]
    assert ck["allreduce_launches"] > 0 and ck["allreduce_us_per_iteration"] > 0 and ck["halo_exchanges"] > 0
    assert out["per_iteration_ms"] > 0 and out["roofline"]["per_iteration_ms"] > 0
    assert "needs distinct devices" in out["rccl_leg"]["skipped"]
    # round 6: the smoke solve before the timed region -- one Gauss-Newton step of 12 PCG iterations through the solver's kernels on every rank, verdict collective
    sm = out["smoke"]
    assert sm["ok"] and sm["comm"] == "peer" and len(sm["ranks"]) == 2 and "fallback" not in sm
    assert all(r["comm_error"] == 0 and r["exception"] is None for r in sm["ranks"]) and sm["ranks"][0]["cost"] == sm["ranks"][1]["cost"]


@pytest.mark.parametrize("world,W,rows_per_rank,ghost,double,onchip_rows", [(2, 600, 64, 8, False, 4), (2, 300, 32, 2, True, 4), (4, 520, 32, 8, False, 8), (2, 260, 64, 4, False, 16), (3, 257, 24, 2, True, 4)])
def test_onchip_linear_solve_across_ranks(world, W, rows_per_rank, ghost, double, onchip_rows):
    """The slab form of the on-chip linear solve (iw_onchip.h, OptAmd_SlabCommExt.onChipPlan): every rank's persistent kernel runs the whole PCG loop on its slab;
    the first / last tile rows of neighbouring ranks hand each other the A p of their edge rows through the edge boxes of the peer window, and the grid-wide sums
    take a rank hop through the mailbox.  Ranks as processes sharing the box's one GPU (their kernels are co-resident: at most 112 / 56 tiles each); result against
    the single-GPU solve, costs bitwise equal between the ranks.  ROWS = 4 / 8 / 16, float and double, middle ranks with a neighbour on both sides."""
    H = world * rows_per_rank
    case = dict(W=W, H=H, double=double, ghost=ghost, kind="gaussNewtonGPU", n=3, l=11, expect_onchip=True, env={"OPT_AMD_ONCHIP_ROWS": str(onchip_rows)})
    P = wl.image_warping(W, H, double=double, random_state=3, mask_fraction=0.06, perturb=0.3)
    c1, x1 = _single(P, case["kind"], nIterations=case["n"], lIterations=case["l"])
    res = _run(world, case)
    tol = 1e-10 if double else 2e-5
    for r in range(world):
        _, costs, unk, row0, rows, mem_kind, err = res[r]
        assert err == 0
        np.testing.assert_allclose(costs, c1, rtol=tol)
        assert costs == res[0][1]
        for a, b in zip(unk, x1):
            assert rel_err(a, b[row0:row0 + rows]) < (1e-9 if double else 2e-5)


@pytest.mark.parametrize("world,W,rows_per_rank,ghost,double,onchip_rows,fail_rank", [(2, 260, 64, 4, False, 16, 1), (2, 300, 32, 2, True, 4, 0), (3, 257, 24, 2, True, 4, 1)])
def test_onchip_across_ranks_one_rank_times_out(world, W, rows_per_rank, ghost, double, onchip_rows, fail_rank):
    """ONE rank's kernel gives up (test hook OPT_AMD_ONCHIP_FAIL_AT on that rank only), the others finish their loops or time out waiting for it.
    The ranks' verdicts are all-reduced on the device before anyone applies its delta, so no rank keeps an update; every rank then redoes the step with the streaming
    slab loop from delta = 0 -- the ROWS = 16 variant accumulates delta in memory while it runs -- and the job ends at the single-GPU result."""
    H = world * rows_per_rank
    case = dict(W=W, H=H, double=double, ghost=ghost, kind="gaussNewtonGPU", n=3, l=11, expect_onchip=True, expect_fallback=True,
                env={"OPT_AMD_ONCHIP_ROWS": str(onchip_rows), "OPT_AMD_ONCHIP_TIMEOUT_MS": "300"}, env_by_rank={fail_rank: {"OPT_AMD_ONCHIP_FAIL_AT": "3"}})
    P = wl.image_warping(W, H, double=double, random_state=3, mask_fraction=0.06, perturb=0.3)
    c1, x1 = _single(P, case["kind"], nIterations=case["n"], lIterations=case["l"])
    res = _run(world, case)
    tol = 1e-10 if double else 2e-5
    for r in range(world):
        _, costs, unk, row0, rows, mem_kind, err = res[r]
        np.testing.assert_allclose(costs, c1, rtol=tol)
        assert costs == res[0][1]
        for a, b in zip(unk, x1):
            assert rel_err(a, b[row0:row0 + rows]) < (1e-9 if double else 2e-5)


def test_onchip_across_ranks_is_all_or_none():
    """One rank's slab does not make whole tiles (33 + 32 rows): that rank cannot run on chip, so nobody does (the decision is an all-reduce) and the streaming slab
    loop runs everywhere."""
    case = dict(W=300, H=65, double=True, ghost=2, kind="gaussNewtonGPU", n=2, l=9, expect_onchip=False, env={"OPT_AMD_ONCHIP_ROWS": "4"})
    P = wl.image_warping(300, 65, double=True, random_state=3, mask_fraction=0.06, perturb=0.3)
    c1, x1 = _single(P, case["kind"], nIterations=case["n"], lIterations=case["l"])
    res = _run(2, case)
    for r in range(2):
        np.testing.assert_allclose(res[r][1], c1, rtol=1e-10)
