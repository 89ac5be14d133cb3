"""GPU parity tests (-m gpu) of the marching PCG iteration template (opt_amd/csrc/stencil_march.h) on its instances: poisson_image_editing (float4 / double4,
Exclude mask), the tests/minimal laplacian (float, no mask) and optical_flow (float2 / double2 with a 2 x 2 block per pixel streamed as operator coefficients).

One launch per PCG iteration (reference loop: solverGPUGaussNewton.t:1056-1092); no A p, no residual vector, delta every second launch -- so the cases walk the
launch-to-launch state machine (1, 2, 3, 4, 5 iterations: first launch, the launch that reads r_0 again, the first rebuilt residual, the first paired delta update,
an odd last launch with a deferred term) on images that are narrower than a wave, exactly one strip, one pixel more, taller / shorter than a workgroup's row
range, with random masks that reach the image border and with no mask at all.  Checked against the CPU oracle: double 1e-10 (cost) / 1e-9 (unknowns), float 1e-5.
"""
import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import assert_close, device_unknowns, flat_unknowns, hip_solver, oracle_solver, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _marching_kernels_only(monkeypatch):
    """Since round 5 images that fit the chip take the on-chip linear solve (stencil_onchip.h, tests/test_onchip_stencil_gpu.py): this module pins the marching loop."""
    monkeypatch.setenv("OPT_AMD_ONCHIP", "0")

SHAPES = [(1, 1), (1, 7), (7, 1), (2, 2), (61, 5), (240, 3), (241, 9), (300, 40), (64, 300), (517, 33)]


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
    """A system with a handful of unknowns is solved exactly after as many iterations as it has distinct eigenvalues (2 for a 2 x 2 image); from there on the cost
    is what cancellation leaves and every further PCG iteration divides round-off by round-off, in the oracle and in the kernel alike.  Tiny images therefore
    take part with the first two launches only (float: the first, whose cost is not yet the converged remainder)."""
    return min(liters, 2 if double else 1) if W * H < 64 else liters


def _pair(oracle_lib, P, nsteps, liters, cost_tol, x_tol):
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
    assert "PCGIteration" in g.kernel_timings() and "PCGStep1" not in g.kernel_timings()
    assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, x_tol, absolute=True, double=P.double)
    g.close(); o.close()


@pytest.mark.parametrize("liters", [1, 2, 3, 4, 5, 12])
@pytest.mark.parametrize("mask", ["random", "none"])
@pytest.mark.parametrize("W,H", SHAPES)
def test_poisson_double(oracle_lib, W, H, mask, liters):
    _pair(oracle_lib, _poisson(W, H, True, W * 3 + H + liters, mask), 2, _cap(W, H, liters), 1e-10, 1e-9)


@pytest.mark.parametrize("liters", [2, 5, 12])
@pytest.mark.parametrize("mask", ["random", "box"])
@pytest.mark.parametrize("W,H", SHAPES)
def test_poisson_float(oracle_lib, W, H, mask, liters):
    _pair(oracle_lib, _poisson(W, H, False, W * 5 + H + liters, mask), 2, _cap(W, H, liters, False), 1e-5, 2e-5)


@pytest.mark.parametrize("liters", [1, 2, 3, 4, 5, 12])
@pytest.mark.parametrize("W,H", SHAPES)
def test_laplacian_float(oracle_lib, W, H, liters):
    _pair(oracle_lib, wl.laplacian(W, H, seed=W + H + liters), 2, _cap(W, H, liters, False), 1e-5, 2e-5)


@pytest.mark.parametrize("liters", [1, 2, 3, 4, 5, 12])
@pytest.mark.parametrize("double", [True, False])
@pytest.mark.parametrize("W,H", [(7, 9), (61, 5), (240, 3), (241, 9), (300, 40), (64, 300), (517, 33)])
def test_optical_flow(oracle_lib, W, H, double, liters):
    """Off-lattice sample positions (seeded initial flow): the operator's per-pixel coefficients are the sampled derivative images at pixel + flow, rebuilt every
    Gauss-Newton step (the second step runs from the first step's flow)."""
    P = wl.optical_flow(W, H, double=double, seed=W + H + liters, init_flow=1.2)
    _pair(oracle_lib, P, 2, liters, 1e-10 if double else 1e-5, 1e-9 if double else 2e-5)


@pytest.mark.parametrize("liters", [1, 2, 3, 4, 5, 12])
@pytest.mark.parametrize("W,H", [(7, 9), (61, 5), (240, 3), (241, 9), (300, 40), (64, 300), (517, 33)])
def test_intrinsic_double(oracle_lib, W, H, liters):
    """intrinsic_image_decomposition: two unknown images (3 + 1 channels per pixel, the template's split layout) and four L_p weights per pixel as operator coefficients.
    The system is ill-conditioned (weights 500 / 1000 / 10000 on differences of ~0.02, unpreconditioned): a 1-ulp difference between libm pow and the device pow is
    amplified ~1e7-fold in a Gauss-Newton step, which is why the double bars are 1e-8 / 1e-7 here as in tests/test_energies_gpu.py; float runs are smoke level there."""
    P = wl.intrinsic_image_decomposition(W, H, double=True, seed=W + H + liters)
    _pair(oracle_lib, P, 2, liters, 1e-8, 1e-7)


def test_intrinsic_march_matches_the_functor_engine_loop(monkeypatch):
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_INTRINSIC_MARCH", on)
        P = wl.intrinsic_image_decomposition(333, 97, double=True, seed=4)
        g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=3, lIterations=10)
        dev = api.to_device(P)
        g.solve(dev)
        assert ("PCGIteration" in g.kernel_timings()) == (on == "1")
        res.append((g.cost(), device_unknowns(P, dev)))
        g.close()
    assert abs(res[0][0] - res[1][0]) <= 1e-8 * abs(res[1][0])
    assert rel_err(res[0][1], res[1][1]) < 1e-7


def test_optical_flow_march_matches_the_functor_engine_loop(monkeypatch):
    res = []
    for on in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_FLOW_MARCH", on)
        P = wl.optical_flow(333, 97, double=True, seed=4, init_flow=0.7)
        g = hip_solver(P, "gaussNewtonGPU", timing=True, nIterations=3, lIterations=25)
        dev = api.to_device(P)
        g.solve(dev)
        assert ("PCGIteration" in g.kernel_timings()) == (on == "1")
        res.append((g.cost(), device_unknowns(P, dev)))
        g.close()
    assert abs(res[0][0] - res[1][0]) <= 1e-10 * abs(res[1][0])
    assert rel_err(res[0][1], res[1][1]) < 1e-9


@pytest.mark.parametrize("double", [False, True])
def test_rows_per_workgroup_do_not_change_the_result_beyond_rounding(monkeypatch, double):
    """OPT_AMD_ITER_ROWS forces the row range of a workgroup: seams fall elsewhere, the per-pixel arithmetic is the same (sums are added in another order)."""
    res = []
    for rows in ("1", "2", "5", "1000"):
        monkeypatch.setenv("OPT_AMD_ITER_ROWS", rows)
        P = _poisson(333, 97, double, 3, "random")
        g = hip_solver(P, "gaussNewtonGPU", nIterations=1, lIterations=9)
        dev = api.to_device(P)
        g.solve(dev)
        res.append((g.cost(), device_unknowns(P, dev)))
        g.close()
    for c, x in res[1:]:
        assert abs(c - res[0][0]) <= (1e-11 if double else 1e-5) * abs(res[0][0])
        assert rel_err(x, res[0][1]) < (1e-10 if double else 1e-5)
