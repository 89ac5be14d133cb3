"""GPU parity tests (-m gpu) of the on-chip linear solve of shape_from_shading (opt_amd/csrc/sfs_onchip.h): Gauss-Newton and Levenberg-Marquardt.

The reference's loop (solverGPUGaussNewton.t:1056-1103: PCGStep1 [+ CtC p], PCGStep2, PCGStep3, the zeta test on Q for LM) runs as ONE persistent launch per outer
step: a wave holds 64 x (R + 4) pixels of p and r in registers and owns the 60 x R in the middle, the A p of the two-pixel ring travels through a tagged image,
five sums per iteration are added by every workgroup in the same order.  Everything here is stepped side by side with the CPU oracle, which follows the
reference's order literally:
  * every kernel variant (R = 4 / 6 / 8 / 10 owned rows per wave x 4 / 8 waves per workgroup; float and double) on small and ragged images with holes (excluded unknowns) and edge masks,
    including images narrower / lower than one tile, one strip + 1 column, rows that end in the middle of a tile;
  * Gauss-Newton (p_0 = r_0 / 4, then z = r: the first alphaNumerator is r_0 . p_0) and LM (CtC, Q with the next iteration's sums, the early-out);
  * LM linear solves with a split residual reset before their last iteration stay on the marching kernels (lIterations > residual_reset_period);
  * several outer steps on one plan (the tags run on), rejected LM steps;
  * the time-out path: nothing is written, the step is redone by the marching kernels, the plan reports it (on_chip_status 2);
  * on-chip against marching kernels on the same input (sums in another order: 1e-12 double);
  * the reference's own input size (640 x 480, examples/shape_from_shading/src/main.cpp:27-38) on the natural variant.
Tolerances: double 1e-10 on costs / 1e-8 on the radius / 1e-9 on unknowns, float 1e-5 on costs (BASELINE.json north_star).
"""
import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import assert_close, device_unknowns, flat_unknowns, hip_solver, oracle_solver, rel_err

pytestmark = pytest.mark.gpu


def _ran_onchip(g):
    return "PCGSolveOnChip" in g.kernel_timings()


def _side_by_side(oracle_lib, P, kind, nsteps, liters, cost_tol, x_tol, radius_tol=None, expect_onchip=True, status=None, later_tol=None, **controls):
    o = oracle_solver(oracle_lib, P, kind, nIterations=nsteps, lIterations=liters, **controls)
    o.set_threads(4)
    g = hip_solver(P, kind, timing=True, nIterations=nsteps, lIterations=liters, **controls)
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    scale = max(abs(o.cost()), 1e-300)
    costs = [(o.cost(), g.cost())]
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b, (a, b, costs)
        costs.append((o.cost(), g.cost()))
        tol = cost_tol if (later_tol is None or len(costs) <= 2) else later_tol
        assert_close("cost" if len(costs) <= 2 else "cost_later", g.cost(), o.cost(), tol, floor=1e-12 * scale, double=P.double, step=len(costs) - 1)
        if radius_tol is not None and (later_tol is None or len(costs) <= 2):
            assert_close("radius", g.trust_region_radius(), o.trust_region_radius(), radius_tol, double=P.double)
        if not a:
            break
    assert _ran_onchip(g) == expect_onchip, g.kernel_timings().keys()
    assert g.on_chip_status() == (status if status is not None else (1 if expect_onchip else 0))
    if x_tol is not None:
        assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, x_tol, absolute=True, double=P.double)
    g.close(); o.close()
    return costs


# one tile; one strip + 1 column; a row count that ends inside a tile; narrower than the ring; a single strip of many tiles; several strips x several tiles
SHAPES = [(40, 32), (61, 9), (130, 37), (5, 70), (123, 4), (200, 150), (3, 3)]
ROWS = [4, 6, 8, 10]


@pytest.mark.parametrize("kind", ["gaussNewtonGPU", "LMGPU"])
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("W,H", SHAPES)
def test_variants_double(oracle_lib, monkeypatch, W, H, rows, waves, kind):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    P = wl.shape_from_shading(W, H, double=True, seed=W + 3 * H + rows, holes=True, noise=2e-3)
    # the first outer step at the contract; the steps after it start from unknowns that differ in their last bits and drive the cost down by orders of magnitude
    # (5 x 70 GN: 2.4 -> 0.087 -> 0.070, 4.5e-10 apart after the third step, the marching kernels likewise): 1e-8
    # (Gauss-Newton on an image a few pixels thin -- 123 x 4, 5 x 70: an undamped, ill-conditioned normal matrix -- turns the last bits of the sums into 3e-9 of the
    # cost after ten iterations; the damped LM step of the same images holds 1e-10)
    thin = min(W, H) <= 5 and kind == "gaussNewtonGPU"
    _side_by_side(oracle_lib, P, kind, 3, 10, 1e-8 if thin else 1e-10, 1e-7, 1e-8 if kind == "LMGPU" else None, later_tol=1e-8)


@pytest.mark.parametrize("kind", ["gaussNewtonGPU", "LMGPU"])
@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("W,H", [(40, 32), (130, 37), (200, 150)])
def test_variants_float(oracle_lib, monkeypatch, W, H, rows, waves, kind):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    P = wl.shape_from_shading(W, H, double=False, seed=W + H + rows, holes=True, noise=2e-3)
    # the first outer step holds the float contract; later steps start from unknowns that already differ in their last bits (see test_onchip_lm_gpu.py)
    _side_by_side(oracle_lib, P, kind, 2, 10, 1e-5, None, 1e-3 if kind == "LMGPU" else None, later_tol=1e-3, **({"q_tolerance": -1e9} if kind == "LMGPU" else {}))


@pytest.mark.parametrize("liters", [1, 2, 3, 7])
@pytest.mark.parametrize("kind", ["gaussNewtonGPU", "LMGPU"])
def test_short_linear_solves(oracle_lib, kind, liters):
    P = wl.shape_from_shading(130, 70, double=True, seed=liters, holes=True, noise=2e-3)
    _side_by_side(oracle_lib, P, kind, 2, liters, 1e-10, 1e-9, 1e-8 if kind == "LMGPU" else None)


@pytest.mark.parametrize("liters,period,onchip", [(10, 10, True), (9, 10, True), (12, 12, True), (12, 5, False), (10, 3, False), (6, 1, False)])
def test_lm_residual_reset_inside_the_solve_stays_on_the_marching_kernels(oracle_lib, liters, period, onchip):
    P = wl.shape_from_shading(72, 56, double=True, seed=2)
    _side_by_side(oracle_lib, P, "LMGPU", 3, liters, 1e-10, 1e-9, 1e-8, expect_onchip=onchip, residual_reset_period=period)


@pytest.mark.parametrize("qtol", [0.5, 0.05, 5.0, 0.0])
def test_lm_q_early_out_double(oracle_lib, qtol):
    """the zeta test decided on chip by every workgroup from the same totals: the same iteration counts as the oracle (1e-10 on the costs says so)"""
    P = wl.shape_from_shading(130, 90, double=True, seed=11, holes=True, noise=2e-3)
    _side_by_side(oracle_lib, P, "LMGPU", 4, 10, 1e-10, 1e-9, 1e-8, q_tolerance=qtol)


def test_lm_rejected_steps(oracle_lib):
    """a trust region far too large: the first steps are rejected (unknowns restored, radius shrinks), the on-chip solve runs again on the same plan"""
    P = wl.shape_from_shading(96, 64, double=True, seed=5, noise=5e-3)
    _side_by_side(oracle_lib, P, "LMGPU", 6, 10, 1e-10, 1e-9, 1e-8, trust_region_radius=1e12)


@pytest.mark.parametrize("kind", ["gaussNewtonGPU", "LMGPU"])
@pytest.mark.parametrize("fail_at", [0, 1, 3])
def test_timeout_path_redoes_the_step_on_the_marching_kernels(oracle_lib, monkeypatch, kind, fail_at):
    monkeypatch.setenv("OPT_AMD_ONCHIP_FAIL_AT", str(fail_at))
    P = wl.shape_from_shading(130, 70, double=True, seed=3, holes=True, noise=2e-3)
    _side_by_side(oracle_lib, P, kind, 3, 10, 1e-10, 1e-9, 1e-8 if kind == "LMGPU" else None, expect_onchip=True, status=2, **({"q_tolerance": -1e9} if kind == "LMGPU" else {}))


@pytest.mark.parametrize("dbl", [True, False])
@pytest.mark.parametrize("kind", ["gaussNewtonGPU", "LMGPU"])
def test_onchip_against_marching_kernels(monkeypatch, kind, dbl):
    P = wl.shape_from_shading(300, 200, double=dbl, seed=8, holes=True, noise=2e-3)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_ONCHIP", flag)
        g = hip_solver(P, kind, timing=True, nIterations=2, lIterations=10)
        dev = api.to_device(P)
        g.init(dev)
        c = [g.cost()]
        g.step(dev); c.append(g.cost())
        assert _ran_onchip(g) == (flag == "1")
        res[flag] = (c, device_unknowns(P, dev))
        g.close()
    tol = 1e-12 if dbl else 1e-5
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=tol)
    assert rel_err(res["1"][1], res["0"][1]) < (1e-11 if dbl else 1e-5)


def test_reference_input_size_double_lm(oracle_lib):
    """640 x 480 (the reference's own input, main.cpp:27-38): the natural variant, two LM steps of 10 PCG iterations against the oracle"""
    P = wl.shape_from_shading(640, 480, double=True, seed=1, holes=True)
    _side_by_side(oracle_lib, P, "LMGPU", 2, 10, 1e-10, 1e-9, 1e-8)
    g = hip_solver(P, "LMGPU")
    d = g.describe()
    g.close()
    assert "on-chip" in d["path"] and d["onchip_rows_per_wave"] == "6" and d["waves_per_workgroup"] == "4", d      # 11 strips x 80 tiles = 880 waves: one per SIMD on 220 CUs
