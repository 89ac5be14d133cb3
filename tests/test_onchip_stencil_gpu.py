"""GPU parity tests (-m gpu) of the on-chip Gauss-Newton linear solve of the 5-point stencil energies (opt_amd/csrc/stencil_onchip.h: march_onchipPcg) on its instances:
poisson_image_editing (float4 / double4, Exclude mask), the tests/minimal laplacian (float, no mask) and optical_flow (float2 / double2, per-pixel operator coefficients).

The reference's loop (solverGPUGaussNewton.t:1056-1092) runs as ONE persistent launch per Gauss-Newton step: a wave holds 64 x (R + 2) pixels of p and r in registers and
owns the 62 x R in the middle, the A p of the one-pixel ring travels through a tagged image, four sums per iteration are added by every workgroup in the same order; the
start is the reference's (p_0 = r_0 / 4, alphaNumerator_0 = r_0 . p_0).  Side by side with the CPU oracle:
  * every kernel variant that keeps its state in registers (R = 2 / 4 / 8 -- pixels of up to 8 bytes: 16 -- rows per wave x 4 / 8 waves per workgroup) on images narrower than a wave, exactly one strip, one
    pixel more, lower than a tile, with random masks that reach the border and with none;
  * 1, 2, 3, 5, 12 PCG iterations, two Gauss-Newton steps on one plan (the tags run on; optical_flow's coefficients are rebuilt);
  * the time-out path (nothing applied, the step redone by the marching kernels, on_chip_status 2) and on-chip against marching kernels on the same input.
Tolerances: double 1e-10 (cost) / 1e-9 (unknowns), float 1e-5.
"""
import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import assert_close, device_unknowns, flat_unknowns, hip_solver, oracle_solver, rel_err

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1), (1, 7), (7, 1), (2, 2), (61, 5), (62, 3), (63, 9), (300, 40), (64, 300), (517, 33)]
VARIANTS = [(2, 4), (4, 4), (8, 4), (2, 8), (4, 8), (8, 8)]


def _poisson(W, H, double, seed, mask):
    P = wl.poisson_image_editing(W, H, double=double, seed=seed)
    rng = np.random.default_rng(seed + 100)
    M = P.params[2]
    if mask == "random":
        M[...] = np.where(rng.random(M.shape) < 0.3, 255.0, 0.0)
    elif mask == "none":
        M[...] = 0.0
    return P


def _cap(W, H, liters, double=True):
    """(tests/test_stencil_march_gpu.py) a system of a handful of unknowns is solved exactly after as many iterations as it has distinct eigenvalues; beyond that every
    PCG iteration divides round-off by round-off, in the oracle and in the kernel alike"""
    return min(liters, 2 if double else 1) if W * H < 64 else liters


def _pair(oracle_lib, P, nsteps, liters, cost_tol, x_tol, status=1):
    o = oracle_solver(oracle_lib, P, "gaussNewtonGPU", nIterations=nsteps, lIterations=liters)
    g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=nsteps, lIterations=liters)
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    scale = max(abs(o.cost()), 1e-300)
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b
        assert_close("cost", g.cost(), o.cost(), cost_tol, floor=1e-9 * scale, double=P.double)
        if not a:
            break
    kt = g.kernel_timings()
    assert "PCGSolveOnChip" in kt and ("PCGIteration" in kt) == (status == 2), kt.keys()
    assert g.on_chip_status() == status
    assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, x_tol, absolute=True, double=P.double)
    g.close(); o.close()


@pytest.mark.parametrize("liters", [1, 2, 5, 12])
@pytest.mark.parametrize("rows,waves", [(2, 4), (4, 4), (8, 4), (2, 8), (4, 8)])      # (8 rows x 8 waves of 32-byte pixels does not fit the registers: not offered)
@pytest.mark.parametrize("mask", ["random", "none"])
@pytest.mark.parametrize("W,H", SHAPES)
def test_poisson_double(oracle_lib, monkeypatch, W, H, mask, rows, waves, liters):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    _pair(oracle_lib, _poisson(W, H, True, W * 3 + H + liters, mask), 2, _cap(W, H, liters), 1e-10, 1e-9)


@pytest.mark.parametrize("liters", [2, 12])
@pytest.mark.parametrize("rows,waves", VARIANTS)
@pytest.mark.parametrize("mask", ["random", "box"])
@pytest.mark.parametrize("W,H", SHAPES)
def test_poisson_float(oracle_lib, monkeypatch, W, H, mask, rows, waves, liters):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    _pair(oracle_lib, _poisson(W, H, False, W * 5 + H + liters, mask), 2, _cap(W, H, liters, False), 1e-5, 2e-5)


@pytest.mark.parametrize("liters", [1, 3, 12])
@pytest.mark.parametrize("rows,waves", VARIANTS + [(16, 4), (16, 8)])
@pytest.mark.parametrize("W,H", SHAPES)
def test_laplacian_float(oracle_lib, monkeypatch, W, H, rows, waves, liters):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    _pair(oracle_lib, wl.laplacian(W, H, seed=W + H + liters), 2, _cap(W, H, liters, False), 1e-5, 2e-5)


# (16 rows per wave are offered for pixels of up to 8 bytes: float2, not double2)
@pytest.mark.parametrize("liters", [1, 3, 12])
@pytest.mark.parametrize("double,rows,waves", [(d, r, w) for d in (True, False) for (r, w) in VARIANTS + [(16, 4), (16, 8)] if not (d and r == 16)])
@pytest.mark.parametrize("W,H", [(7, 9), (61, 5), (62, 3), (63, 9), (300, 40), (64, 300), (517, 33)])
def test_optical_flow(oracle_lib, monkeypatch, W, H, double, rows, waves, liters):
    """off-lattice sample positions (seeded initial flow): the per-pixel coefficients are rebuilt every Gauss-Newton step"""
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    P = wl.optical_flow(W, H, double=double, seed=W + H + liters, init_flow=1.2)
    _pair(oracle_lib, P, 2, liters, 1e-10 if double else 1e-5, 1e-9 if double else 2e-5)


@pytest.mark.parametrize("fail_at", [0, 1, 4])
@pytest.mark.parametrize("energy", ["poisson", "laplacian", "optical_flow"])
def test_timeout_path_redoes_the_step_on_the_marching_kernels(oracle_lib, monkeypatch, energy, fail_at):
    monkeypatch.setenv("OPT_AMD_ONCHIP_FAIL_AT", str(fail_at))
    P = {"poisson": lambda: _poisson(300, 90, True, 5, "random"), "laplacian": lambda: wl.laplacian(300, 90, seed=3),
         "optical_flow": lambda: wl.optical_flow(300, 90, double=True, seed=7, init_flow=1.2)}[energy]()
    dbl = energy != "laplacian"
    _pair(oracle_lib, P, 3, 10, 1e-10 if dbl else 1e-5, 1e-9 if dbl else 2e-5, status=2)


@pytest.mark.parametrize("energy", ["poisson", "optical_flow"])
def test_onchip_against_marching_kernels(monkeypatch, energy):
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_ONCHIP", flag)
        P = _poisson(333, 97, True, 3, "random") if energy == "poisson" else wl.optical_flow(333, 97, double=True, seed=4, init_flow=0.7)
        g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=3, lIterations=10)
        dev = api.to_device(P)
        g.solve(dev)
        assert ("PCGSolveOnChip" in g.kernel_timings()) == (flag == "1")
        res[flag] = (g.cost(), device_unknowns(P, dev))
        g.close()
    assert abs(res["1"][0] - res["0"][0]) <= 1e-11 * abs(res["0"][0])
    assert rel_err(res["1"][1], res["0"][1]) < 1e-10


def test_config1_poisson_256_takes_the_onchip_solve(oracle_lib):
    """BASELINE config 1 (poisson_image_editing 256 x 256 float, GN + 10 PCG iterations)"""
    P = wl.poisson_image_editing(256, 256, double=False, seed=1)
    _pair(oracle_lib, P, 1, 10, 1e-5, 2e-5)
    g = hip_solver(P, "gaussNewtonGPU")
    d = g.describe()
    g.close()
    assert "on-chip" in d["path"], d


# ---- Levenberg-Marquardt (march_onchipPcg<.., LM = true>): CtC, Q with the next iteration's sums, the early-out decided on chip ------------------------------------
def _lm_side_by_side(oracle_lib, P, nsteps, liters, cost_tol, x_tol, radius_tol, expect_onchip=True, status=None, later_tol=None, **controls):
    o = oracle_solver(oracle_lib, P, "LMGPU", nIterations=nsteps, lIterations=liters, **controls)
    o.set_threads(4)
    g = hip_solver(P, "LMGPU", timing=True, nIterations=nsteps, lIterations=liters, **controls)
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
        assert_close("cost" if len(costs) <= 2 else "cost_later", g.cost(), o.cost(), tol, floor=1e-9 * scale, double=P.double, step=len(costs) - 1)
        if later_tol is None or len(costs) <= 2:
            assert_close("radius", g.trust_region_radius(), o.trust_region_radius(), radius_tol, double=P.double)
        if not a:
            break
    assert ("PCGSolveOnChip" in g.kernel_timings()) == expect_onchip, g.kernel_timings().keys()
    assert g.on_chip_status() == (status if status is not None else (1 if expect_onchip else 0))
    if x_tol is not None:
        assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, x_tol, absolute=True, double=P.double)
    g.close(); o.close()


LM_SHAPES = [(7, 9), (61, 5), (62, 3), (63, 9), (300, 40), (64, 300), (517, 33)]


@pytest.mark.parametrize("rows,waves", [(2, 4), (4, 4), (8, 4), (2, 8)])
@pytest.mark.parametrize("mask", ["random", "none"])
@pytest.mark.parametrize("W,H", LM_SHAPES)
def test_lm_poisson_double(oracle_lib, monkeypatch, W, H, mask, rows, waves):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    _lm_side_by_side(oracle_lib, _poisson(W, H, True, W * 3 + H, mask), 3, 10, 1e-10, 1e-9, 1e-8)


@pytest.mark.parametrize("rows,waves", VARIANTS)
@pytest.mark.parametrize("W,H", [(61, 5), (300, 40), (517, 33)])
def test_lm_poisson_float(oracle_lib, monkeypatch, W, H, rows, waves):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    _lm_side_by_side(oracle_lib, _poisson(W, H, False, W * 5 + H, "random"), 2, 10, 1e-5, None, 1e-3, later_tol=1e-3, q_tolerance=-1e9)


@pytest.mark.parametrize("rows,waves", VARIANTS + [(16, 4), (16, 8)])
@pytest.mark.parametrize("W,H", [(61, 5), (300, 40), (64, 300)])
def test_lm_laplacian_float(oracle_lib, monkeypatch, W, H, rows, waves):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    _lm_side_by_side(oracle_lib, wl.laplacian(W, H, seed=W + H), 2, 10, 1e-5, None, 1e-3, later_tol=1e-3, q_tolerance=-1e9)


@pytest.mark.parametrize("rows,waves", [(2, 4), (4, 4), (8, 4), (2, 8), (4, 8)])
@pytest.mark.parametrize("W,H", [(61, 5), (300, 40), (517, 33)])
def test_lm_optical_flow_double(oracle_lib, monkeypatch, W, H, rows, waves):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    _lm_side_by_side(oracle_lib, wl.optical_flow(W, H, double=True, seed=W + H, init_flow=1.2), 3, 10, 1e-10, 1e-9, 1e-8)


@pytest.mark.parametrize("liters,period,onchip", [(10, 10, True), (9, 10, True), (12, 12, True), (12, 5, False), (6, 1, False)])
def test_lm_residual_reset_inside_the_solve_stays_on_the_generic_kernels(oracle_lib, liters, period, onchip):
    _lm_side_by_side(oracle_lib, _poisson(120, 70, True, 9, "random"), 3, liters, 1e-10, 1e-9, 1e-8, expect_onchip=onchip, residual_reset_period=period)


@pytest.mark.parametrize("qtol", [0.5, 0.05, 5.0, 0.0])
def test_lm_q_early_out_double(oracle_lib, qtol):
    _lm_side_by_side(oracle_lib, _poisson(130, 90, True, 11, "random"), 4, 10, 1e-10, 1e-9, 1e-8, q_tolerance=qtol)


@pytest.mark.parametrize("fail_at", [0, 2])
def test_lm_timeout_path(oracle_lib, monkeypatch, fail_at):
    monkeypatch.setenv("OPT_AMD_ONCHIP_FAIL_AT", str(fail_at))
    _lm_side_by_side(oracle_lib, _poisson(130, 70, True, 3, "random"), 3, 10, 1e-10, 1e-9, 1e-8, expect_onchip=True, status=2, q_tolerance=-1e9)


# ---- intrinsic_image_decomposition: two unknown images (3 + 1 channels per pixel: the solver's split layout), four operator coefficients per pixel; the update is
# applied by the solver (the caller's two arrays need not be consecutive) ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("liters", [1, 3, 12])
@pytest.mark.parametrize("rows,waves", [(2, 4), (4, 4), (2, 8)])
@pytest.mark.parametrize("W,H", [(7, 9), (61, 5), (62, 3), (63, 9), (300, 40), (64, 300)])
def test_intrinsic_double(oracle_lib, monkeypatch, W, H, rows, waves, liters):
    """(ill-conditioned, unpreconditioned: 1e-8 / 1e-7 as in tests/test_stencil_march_gpu.py)"""
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    _pair(oracle_lib, wl.intrinsic_image_decomposition(W, H, double=True, seed=W + H + liters), 2, liters, 1e-8, 1e-7)


@pytest.mark.parametrize("rows,waves", VARIANTS)
def test_intrinsic_float_variants_against_the_marching_kernels(monkeypatch, rows, waves):
    """float: the trajectory of this energy is outside the 1e-5 contract for any two implementations (tests/golden/float_envelopes.json); the variants are pinned against the
    marching loop on the same input after one Gauss-Newton step of 5 iterations"""
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_ONCHIP", flag); monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
        P = wl.intrinsic_image_decomposition(200, 120, double=False, seed=4)
        g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=1, lIterations=5)
        dev = api.to_device(P)
        g.solve(dev)
        assert ("PCGSolveOnChip" in g.kernel_timings()) == (flag == "1")
        res[flag] = (g.cost(), device_unknowns(P, dev))
        g.close()
    assert abs(res["1"][0] - res["0"][0]) <= 2e-4 * abs(res["0"][0])
    assert rel_err(res["1"][1], res["0"][1]) < 1e-4


@pytest.mark.parametrize("rows,waves", [(2, 4), (4, 4), (2, 8)])
@pytest.mark.parametrize("W,H", [(61, 5), (300, 40)])
def test_lm_intrinsic_double(oracle_lib, monkeypatch, W, H, rows, waves):
    monkeypatch.setenv("OPT_AMD_ONCHIP_ROWS", str(rows)); monkeypatch.setenv("OPT_AMD_ONCHIP_WAVES", str(waves))
    _lm_side_by_side(oracle_lib, wl.intrinsic_image_decomposition(W, H, double=True, seed=W + H), 3, 10, 1e-8, 1e-6, 1e-6)


def test_intrinsic_timeout_path(oracle_lib, monkeypatch):
    monkeypatch.setenv("OPT_AMD_ONCHIP_FAIL_AT", "1")
    _pair(oracle_lib, wl.intrinsic_image_decomposition(200, 90, double=True, seed=3), 2, 10, 1e-8, 1e-7, status=2)
