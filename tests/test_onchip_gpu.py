"""GPU parity tests (-m gpu) of the on-chip linear solve (opt_amd/csrc/iw_onchip.h): image_warping, Gauss-Newton, unit lattice.

The whole PCG loop of a Gauss-Newton step (reference: solverGPUGaussNewton.t:1056-1092, PCGStep1..3 per iteration) runs as ONE persistent
launch that keeps p, r, A p and delta in registers / LDS and synchronises the grid through tagged 8-byte words.  It is selected by the
solver itself when the image fits (tiles of 256 x 2 ROWS pixels <= CUs); these tests put it against the CPU oracle, the same way the
streaming kernels are tested (tests/test_steady_state_gpu.py):
  * every kernel variant (ROWS = 2 / 4 / 8 / 16 float, 2 / 4 double; A p in registers or LDS, delta in registers or memory) on small and
    ragged images (one tile, several tiles across and down, partial tiles, a single column / row of tiles), with masks;
  * flat and two-level-tree grid sums (bitwise the same result), several groups of 16 workgroups;
  * odd / even / tiny iteration counts (1, 2, 3, 7, 8, 20), two Gauss-Newton steps (the tag counter runs on between launches);
  * the per-iteration scalars against the oracle's trace;
  * the time-out path (a wait gives up -> nothing is applied -> the step is redone by the streaming kernels);
  * the reference's real input sizes (512^2, 640x480) and 1/8 of the metric's image (4096x512) on the natural choice of variant.
Tolerances: double 1e-10 on costs / 1e-9 on unknowns, float 1e-5 (BASELINE.json north_star).
"""
import os

import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import assert_close, device_unknowns, flat_unknowns, hip_solver, oracle_solver, rel_err

pytestmark = pytest.mark.gpu

THREADS = max(1, min(os.cpu_count() or 1, 64))


def _ran_onchip(g):
    return "PCGSolveOnChip" in g.kernel_timings()


def _pair(oracle_lib, P, nsteps, liters, cost_tol, x_tol, expect_onchip=True):
    o = oracle_solver(oracle_lib, P, "gaussNewtonGPU", nIterations=nsteps, lIterations=liters)
    o.set_threads(THREADS if P.params[0].size > 200_000 else 1)
    g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=nsteps, lIterations=liters)
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    scale = max(abs(o.cost()), 1e-300)
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b
        assert_close("cost", g.cost(), o.cost(), cost_tol, floor=1e-12 * scale, double=P.double)
        if not a:
            break
    assert _ran_onchip(g) == expect_onchip, g.kernel_timings().keys()
    if x_tol is not None:
        assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, x_tol, absolute=True, double=P.double)
    g.close(); o.close()


# one tile; tiles across (x) with a partial last tile; tiles down; both; a single pixel column; fewer rows than one wave holds
SHAPES = [(96, 64), (300, 40), (517, 33), (64, 300), (260, 131), (1, 70), (700, 3), (257, 9)]


@pytest.mark.parametrize("liters", [1, 2, 3, 7, 8])
@pytest.mark.parametrize("rows", [2, 4])
@pytest.mark.parametrize("W,H", SHAPES)
def test_variants_double(oracle_lib, monkeypatch, W, H, rows, liters):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows))
    P = wl.image_warping(W, H, double=True, random_state=W * 13 + H + rows + liters, mask_fraction=0.1, perturb=0.3)
    _pair(oracle_lib, P, 2, liters, 1e-10, 1e-9)


@pytest.mark.parametrize("liters", [3, 8, 20])
@pytest.mark.parametrize("rows", [2, 4, 8, 16])
@pytest.mark.parametrize("W,H", SHAPES)
def test_variants_float(oracle_lib, monkeypatch, W, H, rows, liters):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows))
    P = wl.image_warping(W, H, random_state=W * 7 + H + rows + liters, mask_fraction=0.1, perturb=0.3)
    _pair(oracle_lib, P, 2, liters, 1e-5, 2e-5)


@pytest.mark.parametrize("flat", [0, 1000])
@pytest.mark.parametrize("W,H,rows", [(300, 200, 4), (520, 400, 4), (1030, 250, 8), (2050, 130, 16)])
def test_grid_sum_tree_and_flat(oracle_lib, monkeypatch, W, H, rows, flat):
    """50 / 150 / 80 / 45 workgroups: several groups of 16 (the last one partial); the tree (OPT_AMD_ONCHIP_FLAT=0) and the flat sum add in the same order."""
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows))
    monkeypatch.setenv("OPT_AMD_ONCHIP_FLAT", str(flat))
    dbl = rows == 4
    P = wl.image_warping(W, H, double=dbl, random_state=W + H + rows, mask_fraction=0.05, perturb=0.3)
    _pair(oracle_lib, P, 2, 9, 1e-10 if dbl else 1e-5, 1e-9 if dbl else 2e-5)


def test_tree_and_flat_sums_agree_bitwise(monkeypatch):
    res = []
    for flat in (0, 1000):
        monkeypatch.setenv("OPT_AMD_ONCHIP_FLAT", str(flat))
        P = wl.image_warping(520, 400, random_state=5, mask_fraction=0.05, perturb=0.3)
        g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=2, lIterations=12)
        dev = api.to_device(P)
        g.solve(dev)
        assert _ran_onchip(g)
        res.append((g.cost(), device_unknowns(P, dev)))
        g.close()
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("double", [False, True])
def test_onchip_equals_the_streaming_loop(monkeypatch, double):
    """Same iterates as iw_pcgIter2 (the sums are formed from the same terms; only their order over pixels and two horizontal pair evaluations differ)."""
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_ONCHIP", on)
        P = wl.image_warping(600, 300, double=double, random_state=11, mask_fraction=0.05, perturb=0.3)
        g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=2, lIterations=15)
        dev = api.to_device(P)
        g.solve(dev)
        assert _ran_onchip(g) == (on == "1")
        res.append((g.cost(), device_unknowns(P, dev)))
        g.close()
    tol = 1e-11 if double else 2e-5
    assert abs(res[0][0] - res[1][0]) <= tol * abs(res[1][0])
    assert rel_err(res[0][1], res[1][1]) < (1e-10 if double else 2e-5)


def test_trace_against_the_oracle(oracle_lib):
    """alphaNumerator / alphaDenominator / betaNumerator of every iteration (OptAmd_PlanEnableTrace) from the kernel's own sums."""
    P = wl.image_warping(300, 120, double=True, random_state=3, mask_fraction=0.05, perturb=0.3)
    o = oracle_solver(oracle_lib, P, "gaussNewtonGPU", nIterations=1, lIterations=10)
    Pref = P.clone()
    o.solve(Pref.params)
    g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=1, lIterations=10)
    g.enable_trace()
    dev = api.to_device(P)
    g.solve(dev)
    assert _ran_onchip(g)
    to, tg = o.trace(), g.trace()
    assert to.shape == tg.shape and tg.shape[0] == 10
    for col in (2, 3, 4):
        assert np.allclose(tg[:, col], to[:, col], rtol=1e-9, atol=0), (col, tg[:, col], to[:, col])
    g.close(); o.close()


@pytest.mark.parametrize("fail_at", [0, 3, 7])
def test_a_timed_out_wait_leaves_the_unknowns_alone_and_the_step_is_redone(oracle_lib, monkeypatch, capfd, fail_at):
    monkeypatch.setenv("OPT_AMD_ONCHIP_FAIL_AT", str(fail_at))
    P = wl.image_warping(300, 120, double=True, random_state=4, mask_fraction=0.05, perturb=0.3)
    o = oracle_solver(oracle_lib, P, "gaussNewtonGPU", nIterations=3, lIterations=8)
    Pref = P.clone()
    o.solve(Pref.params)
    g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=3, lIterations=8)
    dev = api.to_device(P)
    g.solve(dev)
    t = g.kernel_timings()
    # Tried once, then the streaming loop for the rest of the plan.  (The three steps of this solve are enqueued back to back -- round 6, deferred steps -- so the launches
    # of steps 2 and 3 are already behind the one that times out: they find the sticky failure flag and return at once, nothing of theirs is applied.)
    assert 1 <= t["PCGSolveOnChip"][0] <= 3 and t["PCGIteration"][0] == 3 * 8, t
    assert_close("cost", g.cost(), o.cost(), 1e-10, double=True)
    assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, 1e-9, absolute=True, double=True)
    assert "timed out" in capfd.readouterr().err
    g.close(); o.close()


@pytest.mark.parametrize("W,H,liters", [(512, 512, 10), (640, 480, 10), (1024, 1024, 12), (2048, 1024, 12), (4096, 512, 12)])
def test_natural_variant_float(oracle_lib, W, H, liters):
    """The sizes the path exists for: the reference's own inputs (examples/image_warping/src/main.cpp:98-134: 512^2; the SFS fixture's 640x480), 1024^2 on the
    8-row variant, and 2 M pixels -- 1/8 of the metric's 4096^2 -- on the 16-row variant with A p in LDS and delta in memory."""
    P = wl.image_warping(W, H, random_state=W + H, mask_fraction=0.02, perturb=0.3)
    _pair(oracle_lib, P, 2, liters, 1e-5, 2e-5)


def test_natural_variant_double_512(oracle_lib):
    P = wl.image_warping(512, 512, double=True, random_state=9, mask_fraction=0.02, perturb=0.3)
    _pair(oracle_lib, P, 2, 10, 1e-10, 1e-9)


def test_too_large_or_general_inputs_take_the_streaming_kernels(oracle_lib):
    P = wl.image_warping(2048, 1100, random_state=1, perturb=0.3)              # 2.25 M pixels: does not fit
    _pair(oracle_lib, P, 1, 4, 1e-5, 2e-5, expect_onchip=False)
    P = wl.image_warping(300, 100, double=True, random_state=2, perturb=0.3, jitter_urshape=0.2)      # UrShape is not the unit lattice
    _pair(oracle_lib, P, 1, 4, 1e-10, 1e-9, expect_onchip=False)


def test_many_steps_tag_counter_runs_on(oracle_lib):
    """19 x 8 launches of the example flow's shape on one plan: tags never repeat, the double buffers alternate whatever the parity of the counts."""
    P = wl.image_warping(260, 131, double=True, random_state=21, mask_fraction=0.05, perturb=0.3)
    _pair(oracle_lib, P, 9, 5, 1e-10, 1e-9)


@pytest.mark.parametrize("fail_launch,nsteps", [(0, 4), (2, 6), (3, 12), (7, 12), (9, 12), (5, 6)])
def test_a_time_out_among_deferred_steps_sends_the_solve_back_to_that_step(oracle_lib, monkeypatch, capfd, fail_launch, nsteps):
    """Inside Opt_ProblemSolve the Gauss-Newton steps of image_warping are enqueued back to back (round 6: nothing a step computes steers the next one; costs and on-chip verdicts
    are read every 8th step and at the end).  A wait that times out in the n-th launch leaves that step and every later one unapplied (the flag is sticky: later launches return
    at once, every guarded update is skipped); the host finds WHICH step it was from the per-step words and goes back to it on the streaming kernels.  Same unknowns and costs as
    the oracle, whatever the position of the failure in the window (first step, middle, the step that drains the window, the last step of the solve)."""
    monkeypatch.setenv("OPT_AMD_ONCHIP_FAIL_AT", "3")
    monkeypatch.setenv("OPT_AMD_ONCHIP_FAIL_LAUNCH", str(fail_launch))
    P = wl.image_warping(300, 120, double=True, random_state=4, mask_fraction=0.05, perturb=0.3)
    o = oracle_solver(oracle_lib, P, "gaussNewtonGPU", nIterations=nsteps, lIterations=8)
    Pref = P.clone()
    o.solve(Pref.params)
    g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=nsteps, lIterations=8)
    dev = api.to_device(P)
    g.solve(dev)
    t = g.kernel_timings()
    assert "PCGIteration" in t and "PCGSolveOnChip" in t, t.keys()
    assert_close("cost", g.cost(), o.cost(), 1e-10, double=True)
    assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, 1e-9, absolute=True, double=True)
    assert "timed out" in capfd.readouterr().err
    g.close(); o.close()


def test_deferred_steps_print_every_cost_in_order(oracle_lib, capfd):
    """verbosity > 0: the "cost: a -> b" lines of the deferred steps come out in order when they are settled, with the values step by step gives."""
    import ctypes
    P = wl.image_warping(300, 120, double=True, random_state=5, mask_fraction=0.05, perturb=0.3)
    outs = []
    for whole in (False, True):
        g = hip_solver(P, "gaussNewtonGPU", verbosity=1, nIterations=11, lIterations=6)
        dev = api.to_device(P)
        ctypes.CDLL(None).fflush(None); capfd.readouterr()
        if whole:
            g.solve(dev)
        else:
            g.init(dev)
            while g.step(dev):
                pass
        ctypes.CDLL(None).fflush(None)
        outs.append(([ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("cost:") or ln.startswith("final cost")], g.cost(), device_unknowns(P, dev)))
        g.close()
    assert len(outs[0][0]) == 12 and outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1] and np.array_equal(outs[0][2], outs[1][2])
