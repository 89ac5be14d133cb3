"""CPU tests (-m "not gpu") of the multi-GPU host logic: slab layout / split / merge, and -- with two gloo
processes -- a distributed PCG step built from the same layout + halo/all-reduce protocol the GPU path uses,
with the CPU oracle doing each rank's local J^T J on its slab (owned rows + ghost rows).  The result must equal
the single-process oracle on the whole image."""
import os
import socket

import numpy as np
import pytest

from opt_amd import slab, workloads as wl
from helpers import flat_unknowns


def test_layout_partitions_rows():
    for H, world in [(4096, 8), (53, 3), (7, 7), (10, 4)]:
        lays = [slab.SlabLayout(64, H, r, world, ghost=1 if H < 2 * world else 2) for r in range(world)]
        assert lays[0].row0 == 0 and lays[-1].row0 + lays[-1].rows == H
        for a, b in zip(lays, lays[1:]):
            assert a.row0 + a.rows == b.row0
        assert max(l.rows for l in lays) - min(l.rows for l in lays) <= 1
    with pytest.raises(ValueError):
        slab.SlabLayout(8, 3, 0, 4)


def test_split_and_merge_roundtrip():
    P = wl.image_warping(20, 13, random_state=1, mask_fraction=0.1, perturb=0.2)
    world = 3
    lays = [slab.SlabLayout(20, 13, r, world, ghost=1) for r in range(world)]
    locs = [slab.split_problem(P, l) for l in lays]
    for l, q in zip(lays, locs):
        assert q.dims == (20, l.rows + 2)
        np.testing.assert_array_equal(q.params[0][1:-1], P.params[0][l.owned])      # owned rows
        if l.has_up():
            np.testing.assert_array_equal(q.params[2][0], P.params[2][l.row0 - 1])  # ghost row above = neighbour's last row
        else:
            assert np.all(q.params[4][0] == 255)                                    # outside the image: masked
        if l.has_down():
            np.testing.assert_array_equal(q.params[2][-1], P.params[2][l.row0 + l.rows])
        else:
            assert np.all(q.params[4][-1] == 255)
        assert q.params[5].shape == ()                                             # scalars shared
    Q = P.clone()
    for s in Q.unknown_slots:
        Q.params[s][...] = 0
    slab.merge_unknowns(Q, lays, [[q.params[s] for s in q.unknown_slots] for q in locs])
    for s in P.unknown_slots:
        np.testing.assert_array_equal(Q.params[s], P.params[s])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.binding import OracleSolver
    W, H = 18, 14
    P = wl.image_warping(W, H, double=True, random_state=2, mask_fraction=0.08, perturb=0.3)
    lay = slab.SlabLayout(W, H, rank, world, ghost=1)
    loc = slab.split_problem(P, lay)
    o = OracleSolver("image_warping", "gaussNewtonGPU", True, loc.dims)
    npx = W * lay.local_H

    def owned_mask():
        m = np.zeros((lay.local_H, W), dtype=bool); m[1:-1] = True
        m &= (loc.params[4] == 0)          # excluded (masked) pixels are not unknowns (solver.t:371)
        return np.concatenate([np.repeat(m.reshape(-1), 2), m.reshape(-1)])

    def halo(vec):   # exchange one row per unknown image with the slab neighbours (what OptAmd_SlabComm.haloExchange does)
        imgs = [vec[:2 * npx].reshape(lay.local_H, W, 2), vec[2 * npx:].reshape(lay.local_H, W)]
        for im in imgs:
            reqs = []
            up, down = torch.from_numpy(im[1].copy()), torch.from_numpy(im[-2].copy())
            rup, rdown = torch.zeros_like(up), torch.zeros_like(down)
            if lay.has_up():
                reqs += [dist.isend(up, rank - 1), dist.irecv(rup, rank - 1)]
            if lay.has_down():
                reqs += [dist.isend(down, rank + 1), dist.irecv(rdown, rank + 1)]
            for r in reqs:
                r.wait()
            if lay.has_up():
                im[0] = rup.numpy()
            if lay.has_down():
                im[-1] = rdown.numpy()

    def allsum(x):
        t = torch.tensor([x], dtype=torch.float64); dist.all_reduce(t); return float(t.item())

    own = owned_mask()
    # distributed PCG (solverGPUGaussNewton.t:1032-1091) on slabs: r = -J^T F, Jacobi preconditioner, 6 iterations
    f, d = o.eval_jtf(loc.params)
    r = np.where(own, -f, 0.0); pre = np.where(own, 1.0 / (1.0 + np.sqrt(d)) ** 2, 0.0)
    p = pre * r; delta = np.zeros_like(p)
    aNum = allsum(float(r @ p))
    for _ in range(6):
        halo(p)
        Ap = np.where(own, o.apply_jtj(loc.params, p), 0.0)
        aDen = allsum(float((p * own) @ Ap))
        alpha = aNum / aDen if aDen > 0 else 0.0
        delta += alpha * p * own; r -= alpha * Ap
        z = pre * r
        bNum = allsum(float(z @ r))
        beta = bNum / aNum if aNum > 0 else 0.0
        p = z + beta * p * own
        aNum = bNum
    q.put((rank, lay.row0, lay.rows, delta[:2 * npx].reshape(lay.local_H, W, 2)[1:-1].copy(), delta[2 * npx:].reshape(lay.local_H, W)[1:-1].copy()))
    dist.destroy_process_group()


def test_two_rank_gloo_pcg_matches_single_process(oracle_lib):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: the oracle's own PCG on the whole image
    W, H = 18, 14
    P = wl.image_warping(W, H, double=True, random_state=2, mask_fraction=0.08, perturb=0.3)
    o = oracle_lib.OracleSolver("image_warping", "gaussNewtonGPU", True, P.dims)
    o.set("nIterations", 1); o.set("lIterations", 6)
    o.init(P.params); o.step(P.params)
    delta = o.vector("delta")
    dO, dA = delta[:2 * W * H].reshape(H, W, 2), delta[2 * W * H:].reshape(H, W)
    for rank, row0, rows, lo, la in parts:
        np.testing.assert_allclose(lo, dO[row0:row0 + rows], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(la, dA[row0:row0 + rows], rtol=1e-9, atol=1e-12)


def test_split_with_two_ghost_rows():
    """ghost=2 (the default: lets image_warping iterate without A*p in memory): two rows of the neighbours either side,
    rows outside the global image zero-filled and masked out, merge takes the owned rows only."""
    P = wl.image_warping(12, 11, random_state=4, mask_fraction=0.0)
    lays = [slab.SlabLayout(12, 11, r, 3) for r in range(3)]
    assert all(l.ghost == 2 and l.local_H == l.rows + 4 for l in lays)
    locs = [slab.split_problem(P, l) for l in lays]
    for l, q in zip(lays, locs):
        np.testing.assert_array_equal(q.params[0][2:-2], P.params[0][l.owned])
        if l.has_up():
            np.testing.assert_array_equal(q.params[2][:2], P.params[2][l.row0 - 2:l.row0])
        else:
            assert (q.params[4][:2] != 0).all() and (q.params[0][:2] == 0).all()
        if l.has_down():
            np.testing.assert_array_equal(q.params[2][-2:], P.params[2][l.row0 + l.rows:l.row0 + l.rows + 2])
        else:
            assert (q.params[4][-2:] != 0).all()
    Q = P.clone()
    for s_ in Q.unknown_slots:
        Q.params[s_][...] = -1
    slab.merge_unknowns(Q, lays, [[q.params[s_] for s_ in P.unknown_slots] for q in locs])
    for s_ in P.unknown_slots:
        np.testing.assert_array_equal(Q.params[s_], P.params[s_])
    with pytest.raises(ValueError):
        slab.SlabLayout(8, 7, 0, 4)          # 4 slabs x 2 ghost rows need at least 8 image rows


def _rank_main_two_ghost(rank, world, port, q, G=2):
    """The protocol of the A*p-free iteration on slabs (OptAmd_PlanSetSlab with G >= 2 ghost rows): no A*p vector crosses ranks; the G
    edge rows of r_k and p_k do, once every G - 1 iterations.  With r and p valid v rows out, J^T J p is valid v - 1 rows out, so a
    rank updates r and p on its own rows plus G - j ghost rows in iteration j of a period (none in the last: they are about to be
    overwritten), and sums dot products over own rows only."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.binding import OracleSolver
    W, H = 18, 14
    P = wl.image_warping(W, H, double=True, random_state=2, mask_fraction=0.08, perturb=0.3)
    lay = slab.SlabLayout(W, H, rank, world, ghost=G)
    loc = slab.split_problem(P, lay)
    o = OracleSolver("image_warping", "gaussNewtonGPU", True, loc.dims)
    LH = lay.local_H
    npx = W * LH

    def rows_mask(lo, hi):   # unknown-vector mask of local rows [lo, hi) that are not excluded
        m = np.zeros((LH, W), dtype=bool); m[lo:hi] = True
        m &= (loc.params[4] == 0)
        return np.concatenate([np.repeat(m.reshape(-1), 2), m.reshape(-1)])

    own = rows_mask(G, LH - G)

    def exchange(vec):       # the neighbours' two edge rows -> my two ghost rows, per unknown image
        for im in (vec[:2 * npx].reshape(LH, W, 2), vec[2 * npx:].reshape(LH, W)):
            reqs = []
            up, down = torch.from_numpy(im[G:2 * G].copy()), torch.from_numpy(im[LH - 2 * G:LH - G].copy())
            rup, rdown = torch.zeros_like(up), torch.zeros_like(down)
            if lay.has_up():
                reqs += [dist.isend(up, rank - 1), dist.irecv(rup, rank - 1)]
            if lay.has_down():
                reqs += [dist.isend(down, rank + 1), dist.irecv(rdown, rank + 1)]
            for r_ in reqs:
                r_.wait()
            if lay.has_up():
                im[:G] = rup.numpy()
            if lay.has_down():
                im[LH - G:] = rdown.numpy()

    def allsum(x):
        t = torch.tensor([x], dtype=torch.float64); dist.all_reduce(t); return float(t.item())

    f, d = o.eval_jtf(loc.params)                                   # valid on own rows (their residuals only reach the first ghost row)
    r = np.where(own, -f, 0.0); pre = np.where(own, 1.0 / (1.0 + np.sqrt(d)) ** 2, 0.0)
    exchange(r); exchange(pre)                                      # once per Gauss-Newton step
    p = pre * r; delta = np.zeros_like(p)
    aNum = allsum(float((r * own) @ p))
    period, j, exchanges = max(1, G - 1), 0, 0
    for _ in range(6):
        j += 1
        due = j >= period
        ext = 0 if due else G - j                                   # ghost rows this iteration keeps current by itself
        upd = rows_mask(G - ext, LH - G + ext)
        Ap = np.where(upd, o.apply_jtj(loc.params, p), 0.0)         # p is valid ext + 1 rows out (at least), so A p is valid ext rows out
        aDen = allsum(float((p * own) @ Ap))
        alpha = aNum / aDen if aDen > 0 else 0.0
        delta += alpha * p * own
        r = np.where(upd, r - alpha * Ap, r)
        z = pre * r
        bNum = allsum(float((z * own) @ r))
        beta = bNum / aNum if aNum > 0 else 0.0
        p = np.where(upd, z + beta * p, p)
        aNum = bNum
        if due:
            exchange(r); exchange(p); j = 0; exchanges += 1         # refresh all G ghost rows for the next period
    assert exchanges == 6 // period
    q.put((rank, lay.row0, lay.rows, delta[:2 * npx].reshape(LH, W, 2)[G:LH - G].copy(), delta[2 * npx:].reshape(LH, W)[G:LH - G].copy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("ghost", [2, 4])      # 4: the neighbours' rows cross every third iteration only
def test_two_rank_gloo_two_ghost_row_protocol(oracle_lib, ghost):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main_two_ghost, args=(r, world, port, q, ghost)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    W, H = 18, 14
    P = wl.image_warping(W, H, double=True, random_state=2, mask_fraction=0.08, perturb=0.3)
    o = oracle_lib.OracleSolver("image_warping", "gaussNewtonGPU", True, P.dims)
    o.set("nIterations", 1); o.set("lIterations", 6)
    o.init(P.params); o.step(P.params)
    delta = o.vector("delta")
    dO, dA = delta[:2 * W * H].reshape(H, W, 2), delta[2 * W * H:].reshape(H, W)
    for rank, row0, rows, lo, la in parts:
        np.testing.assert_allclose(lo, dO[row0:row0 + rows], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(la, dA[row0:row0 + rows], rtol=1e-9, atol=1e-12)


def test_bench_launcher_spawns_the_requested_ranks():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run with 2 ranks (--cpu-smoke: the ranks
    rendezvous over gloo and report what they saw, no GPU needed); a launcher whose WORLD_SIZE disagrees with --gpus is an error."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-smoke"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2
    env["WORLD_SIZE"] = "1"
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-smoke"], env=env, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "must agree" in bad.stderr


@pytest.mark.parametrize("W,H,world,ghost", [(40, 64, 8, 8), (33, 50, 3, 2), (16, 16, 1, 4), (24, 41, 4, 5)])
def test_rank_local_workload_equals_the_slab_of_the_global_one(W, H, world, ghost):
    """bench.py's ranks build only their own rows (workloads.image_warping_rows; 8192^2 would be 2 GB of host arrays per rank otherwise):
    bit-identical to split_problem of the global problem, for every rank, float and double."""
    from opt_amd import slab, workloads as wl
    for dbl in (False, True):
        P = wl.image_warping(W, H, double=dbl)
        for r in range(world):
            lay = slab.SlabLayout(W, H, r, world, ghost)
            a = slab.split_problem(P, lay)
            b = wl.image_warping_rows(W, H, lay.row0 - ghost, lay.row0 + lay.rows + ghost, double=dbl)
            assert a.dims == b.dims and a.unknown_slots == b.unknown_slots
            for x, y in zip(a.params, b.params):
                assert np.asarray(x).dtype == np.asarray(y).dtype
                np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
