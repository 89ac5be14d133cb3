"""GPU parity tests (-m gpu) of the persistent ARAP linear solve (opt_amd/csrc/arap_onchip.h): arap_mesh_deformation, Gauss-Newton, float, symmetric graphs.

The whole PCG loop of a Gauss-Newton step (reference: solverGPUGaussNewton.t:1056-1092 with the graph kernels :687-706) runs as ONE persistent launch: a thread owns
up to four vertices (p, r, A p in registers), publishes its search direction once per iteration as tagged 16-byte pieces its neighbours poll, and the grid meets once
per iteration for the four sums.  The solver selects it by itself for float plans on symmetric graphs of up to 2048 vertices per CU; these tests put it against the
CPU oracle the way tests/test_onchip_gpu.py tests image_warping's:
  * every kernel variant (1 / 2 / 4 vertices per thread: up to 131 k / 262 k / 524 k vertices on 256 CUs) incl. BASELINE config 4's 708 x 707 mesh;
  * tiny and ragged meshes (fewer vertices than one workgroup, vertex counts that are no multiple of anything), odd / even / tiny iteration counts;
  * several Gauss-Newton steps on one plan (the tag counter runs on), the per-iteration scalars against the oracle's trace;
  * the time-out path (nothing applied -> the step is redone by the two-kernel loop); same iterates as the two-kernel loop; double / LM / asymmetric graphs stay off it.
Tolerance: the float contract, 1e-5 on costs (BASELINE.json north_star).
"""
import os

import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import device_unknowns, flat_unknowns, hip_solver, oracle_solver, rel_err

pytestmark = pytest.mark.gpu


def _ran_onchip(g):
    return "PCGSolveOnChip" in g.kernel_timings()


def _pair(oracle_lib, P, nsteps, liters, cost_tol, x_tol, expect_onchip=True, kind="gaussNewtonGPU"):
    o = oracle_solver(oracle_lib, P, kind, nIterations=nsteps, lIterations=liters)
    g = hip_solver(P, kind, timing=True, nIterations=nsteps, lIterations=liters)
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    scale = max(abs(o.cost()), 1e-300)
    costs = [(o.cost(), g.cost())]
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b
        costs.append((o.cost(), g.cost()))
        assert abs(g.cost() - o.cost()) <= cost_tol * max(abs(o.cost()), 1e-12 * scale), costs
        if not a:
            break
    assert _ran_onchip(g) == expect_onchip, g.kernel_timings().keys()
    assert g.on_chip_status() == (1 if expect_onchip else 0)
    if x_tol is not None:
        assert rel_err(device_unknowns(P, dev), flat_unknowns(Pref)) < x_tol
    g.close(); o.close()


@pytest.mark.parametrize("liters", [1, 2, 3, 8, 25])
@pytest.mark.parametrize("nx,ny", [(12, 9), (23, 17), (41, 29), (100, 83), (300, 257)])
def test_small_and_ragged_meshes(oracle_lib, nx, ny, liters):
    """108 ... 77 k vertices: less than one workgroup, a handful of workgroups with a ragged tail, a grid whose last workgroups own nothing."""
    P = wl.arap_mesh_deformation(nx, ny, seed=nx + ny + liters, perturb=0.01)
    _pair(oracle_lib, P, 2, liters, 1e-5, 2e-5)


@pytest.mark.parametrize("nx,ny,liters", [(400, 390, 12), (512, 500, 8)])
def test_two_vertices_per_thread(oracle_lib, nx, ny, liters):
    """156 k / 256 k vertices: the VPT = 2 variant."""
    P = wl.arap_mesh_deformation(nx, ny, seed=3, perturb=0.01)
    _pair(oracle_lib, P, 1, liters, 1e-5, 2e-5)


def test_config4_mesh_four_vertices_per_thread(oracle_lib):
    """BASELINE config 4: 708 x 707 = 500 556 vertices, 3.0 M half-edges: the VPT = 4 variant on the full chip."""
    P = wl.arap_mesh_deformation(708, 707, perturb=0.01)
    _pair(oracle_lib, P, 1, 10, 1e-5, 2e-5)


def test_many_steps_tag_counter_runs_on(oracle_lib):
    P = wl.arap_mesh_deformation(60, 50, seed=21, perturb=0.01)
    _pair(oracle_lib, P, 9, 5, 1e-5, 2e-5)


def test_trace_against_the_oracle(oracle_lib):
    """alphaNumerator / alphaDenominator / betaNumerator of every iteration from the kernel's own sums."""
    P = wl.arap_mesh_deformation(60, 50, seed=3, perturb=0.01)
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
        assert np.allclose(tg[:, col], to[:, col], rtol=2e-4, atol=0), (col, tg[:, col], to[:, col])
    g.close(); o.close()


def test_onchip_equals_the_two_kernel_loop(monkeypatch):
    """The same arithmetic per half-edge pair as arap_applySym / arap_flatStepRec; only the order of the sums over vertices differs."""
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_ONCHIP", on)
        P = wl.arap_mesh_deformation(200, 190, seed=11, perturb=0.01)
        g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=2, lIterations=30)
        dev = api.to_device(P)
        g.solve(dev)
        assert _ran_onchip(g) == (on == "1")
        res.append((g.cost(), device_unknowns(P, dev)))
        g.close()
    assert abs(res[0][0] - res[1][0]) <= 2e-5 * abs(res[1][0])
    assert rel_err(res[0][1], res[1][1]) < 2e-5


@pytest.mark.parametrize("fail_at", [0, 3, 7])
def test_a_timed_out_wait_leaves_the_unknowns_alone_and_the_step_is_redone(oracle_lib, monkeypatch, capfd, fail_at):
    monkeypatch.setenv("OPT_AMD_ONCHIP_FAIL_AT", str(fail_at))
    P = wl.arap_mesh_deformation(60, 50, seed=4, perturb=0.01)
    o = oracle_solver(oracle_lib, P, "gaussNewtonGPU", nIterations=3, lIterations=8)
    Pref = P.clone()
    o.solve(Pref.params)
    g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=3, lIterations=8)
    dev = api.to_device(P)
    g.solve(dev)
    t = g.kernel_timings()
    assert t["PCGSolveOnChip"][0] == 1 and "PCGStep2+PCGStep3" in t          # tried once, then the two-kernel loop for the rest of the plan
    assert g.on_chip_status() == 2
    assert abs(g.cost() - o.cost()) <= 1e-5 * abs(o.cost())
    assert rel_err(device_unknowns(P, dev), flat_unknowns(Pref)) < 2e-5
    assert "timed out" in capfd.readouterr().err
    g.close(); o.close()


def test_double_lm_and_asymmetric_graphs_keep_the_launch_per_iteration_loops(oracle_lib):
    P = wl.arap_mesh_deformation(41, 29, double=True, seed=5, perturb=0.01)
    _pair(oracle_lib, P, 1, 6, 1e-10, 1e-9, expect_onchip=False)
    P = wl.arap_mesh_deformation(41, 29, seed=5, perturb=0.01)
    _pair(oracle_lib, P, 2, 6, 1e-5, None, expect_onchip=False, kind="LMGPU")
    P = wl.arap_mesh_deformation(23, 17, seed=3, perturb=0.01)
    keep = np.ones(P.meta["n_edges"], dtype=bool); keep[np.arange(5, len(keep), 7)] = False      # every seventh half-edge dropped: most of them leave their reverse behind
    P.params[7] = np.ascontiguousarray(P.params[7][keep]); P.params[8] = np.ascontiguousarray(P.params[8][keep]); P.params[6] = np.array(int(keep.sum()), dtype=np.int32)
    P.meta["n_edges"] = int(keep.sum())
    _pair(oracle_lib, P, 1, 6, 1e-5, 2e-5, expect_onchip=False)
