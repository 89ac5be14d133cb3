"""GPU parity tests (-m gpu) of the row-marching kernels in the regime the benchmark runs them in.

The single-kernel PCG iteration (`iw_pcgIter2`, energy_image_warping.hip) sizes its grid to one co-resident wave of workgroups, so at
4096^2 a workgroup marches ~98 rows (three trips per loop pass with by-name register rotation, a 3-deep prefetch, a row-triple
barrier, alternating sweep direction, paired delta updates), while a small test image gives every workgroup one or two rows.  These tests
put that steady state against the CPU oracle (reference sequencing: solverGPUGaussNewton.t:421-550, 1016-1177):
  * large images on the natural grid (1536x1024 double: 13 rows per group; 2048^2 float: 25);
  * small and ragged images with OPT_AMD_ITER_ROWS forcing 3 / 4 / 5 / 7 / 98 rows per workgroup;
  * odd and even launch counts (paired delta + pcgFinish, both sweep directions), Levenberg-Marquardt, general UrShape;
  * the frozen oracle trajectory of the benchmark workload itself (tests/golden/bench_costs.json, 400 PCG iterations per step);
  * one full-size step of BASELINE configs 3 (SFS 1024^2 double LM) and 4 (ARAP 500 k vertices) against the oracle.
Tolerances: double 1e-10 on costs / 1e-9 on unknowns, float 1e-5 on costs (BASELINE.json north_star).
"""
import json
import os
import sys

import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import assert_close, device_unknowns, flat_unknowns, hip_solver, oracle_solver, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _streaming_kernels_under_test(monkeypatch):
    """These tests exercise the streaming (one launch per PCG iteration) kernels; small unit-lattice images would otherwise take the on-chip
    linear solve (iw_onchip.h, tests/test_onchip_gpu.py)."""
    monkeypatch.setenv("OPT_AMD_ONCHIP", "0")

THREADS = max(1, min(os.cpu_count() or 1, 64))
HERE = os.path.dirname(os.path.abspath(__file__))


def _pair(oracle_lib, P, kind, nsteps, liters, cost_tol, x_tol, radius_tol=None, **extra):
    """Step the oracle and the HIP solver side by side on the same inputs; compare cost (and LM radius) after every step."""
    o = oracle_solver(oracle_lib, P, kind, nIterations=nsteps, lIterations=liters, **extra)
    o.set_threads(THREADS if P.params[0].size > 200_000 else 1)
    g = hip_solver(P, kind, nIterations=nsteps, lIterations=liters, **extra)
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    scale = max(abs(o.cost()), 1e-300)
    assert_close("cost0", g.cost(), o.cost(), cost_tol, floor=scale, double=P.double)
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b
        assert_close("cost", g.cost(), o.cost(), cost_tol, floor=1e-12 * scale, double=P.double)
        if radius_tol is not None:
            assert_close("radius", g.trust_region_radius(), o.trust_region_radius(), radius_tol, double=P.double)
        if not a:
            break
    if x_tol is not None:
        assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, x_tol, absolute=True, double=P.double)
    g.close(); o.close()


# ---- (a) large images, natural grid, double ------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H,liters", [(1536, 1024, 7), (1536, 1024, 8), (1536, 1024, 20), (1441, 600, 8), (719, 1000, 7), (721, 1200, 20)])
def test_large_image_double_natural_grid(oracle_lib, W, H, liters):
    """>= 6 rows per workgroup on the grid the solver picks itself; odd and even launch counts (the deferred delta term is flushed
    by pcgFinish after an odd one), strips 720 k +- 1 wide."""
    P = wl.image_warping(W, H, double=True, random_state=W + 7 * H + liters, mask_fraction=0.05, perturb=0.3)
    _pair(oracle_lib, P, "gaussNewtonGPU", 1, liters, 1e-10, 1e-9)


# ---- (b) forced rows per workgroup on small / ragged images ----------------------------------------------------------------------
FORCED = [(61, 301), (721, 205), (1441, 103), (59, 98), (120, 197), (64, 99), (7, 400)]


@pytest.mark.parametrize("liters", [7, 8])
@pytest.mark.parametrize("rows", [3, 4, 5, 7, 98])
@pytest.mark.parametrize("W,H", FORCED)
def test_forced_rows_per_group_double(oracle_lib, monkeypatch, W, H, rows, liters):
    monkeypatch.setenv("OPT_AMD_ITER_ROWS", str(rows))
    P = wl.image_warping(W, H, double=True, random_state=W * 31 + H + rows, mask_fraction=0.1, perturb=0.3)
    _pair(oracle_lib, P, "gaussNewtonGPU", 2, liters, 1e-10, 1e-9)


@pytest.mark.parametrize("rows", [5, 98])
@pytest.mark.parametrize("jitter,kind", [(0.0, "LMGPU"), (0.2, "gaussNewtonGPU"), (0.2, "LMGPU")])
def test_forced_rows_lm_and_general_urshape(oracle_lib, monkeypatch, jitter, kind, rows):
    """Levenberg-Marquardt (CtC, Q sums, the residual reset every 10th iteration and the restart launch after it) and the general-UrShape
    kernel (PRE == 2, U read per pixel) with many rows per workgroup."""
    monkeypatch.setenv("OPT_AMD_ITER_ROWS", str(rows))
    P = wl.image_warping(305, 250, double=True, random_state=17 + rows, mask_fraction=0.05, perturb=0.3, jitter_urshape=jitter)
    _pair(oracle_lib, P, kind, 2, 25, 1e-10, 1e-9, radius_tol=1e-8 if kind == "LMGPU" else None)


@pytest.mark.parametrize("jitter,kind", [(0.0, "LMGPU"), (0.2, "gaussNewtonGPU")])
def test_large_image_lm_and_general_urshape_natural_grid(oracle_lib, jitter, kind):
    P = wl.image_warping(1441, 700, double=True, random_state=29, mask_fraction=0.05, perturb=0.3, jitter_urshape=jitter)
    _pair(oracle_lib, P, kind, 1, 21, 1e-10, 1e-9, radius_tol=1e-8 if kind == "LMGPU" else None)


# ---- (c) float at BASELINE config 2's size -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows", [None, 98])
def test_2048_float_20_iterations(oracle_lib, monkeypatch, rows):
    """image_warping 2048^2 float (BASELINE config 2), 1 GN x 20 PCG, cost at the 1e-5 bar: natural grid (25 rows per workgroup) and the
    benchmark's 98."""
    if rows:
        monkeypatch.setenv("OPT_AMD_ITER_ROWS", str(rows))
    P = wl.image_warping(2048, 2048)
    rng = np.random.default_rng(2)                                 # a rougher start than the plain benchmark input: perturbed offsets and angles
    P.params[0] += (0.3 * rng.standard_normal(P.params[0].shape)).astype(np.float32)
    P.params[1] += (0.1 * rng.standard_normal(P.params[1].shape)).astype(np.float32)
    _pair(oracle_lib, P, "gaussNewtonGPU", 1, 20, 1e-5, 1e-5)


def test_2048_float_lm_25_iterations(oracle_lib):
    """The same image under Levenberg-Marquardt on the launch-per-iteration loop of round 6 (no residual vector, Q by the CG recurrence, paired delta): two outer
    steps of 25 PCG iterations -- two split residual resets with their restart launches and re-anchored Q, an odd number of launches (the tail adds an owed delta term) -- at the
    size the reference's LM-only large-image mode is for (examples/image_warping/src/main.cpp:121-129), natural grid: costs and the trust-region radius at the float bars."""
    P = wl.image_warping(2048, 2048)
    rng = np.random.default_rng(2)
    P.params[0] += (0.3 * rng.standard_normal(P.params[0].shape)).astype(np.float32)
    P.params[1] += (0.1 * rng.standard_normal(P.params[1].shape)).astype(np.float32)
    _pair(oracle_lib, P, "LMGPU", 2, 25, 1e-5, 1e-5, radius_tol=1e-3)


# ---- (d) the benchmark workload against its frozen oracle trajectory ----------------------------------------------------------------
def _golden():
    with open(os.path.join(HERE, "golden", "bench_costs.json")) as f:
        return json.load(f)


def _spread():
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import reference_spread as rs
    return rs


@pytest.mark.parametrize("size", [2048, 4096])
def test_benchmark_workload_against_frozen_oracle_costs(size):
    """bench.py's exact workload (400 PCG iterations per Gauss-Newton step) against the frozen exact-order oracle trajectory (tests/golden/bench_costs.json).  Over 400
    PCG iterations on this ill-conditioned system no two legal runs of the reference's own arithmetic stay within 1e-5 of each other: its per-warp float atomics commit in
    an undefined order, and the frozen runs of the oracle's reference-order mode (tests/golden/reference_order_costs*.json; tools/reference_spread.py) end 2.3e-3 / 1.4e-3
    apart after step 1 / 2 at 2048^2 and 8.1e-3 / 1.9e-3 at 4096^2.  The 1e-5 contract is checked where it is meaningful (<= 20 iterations, the tests above); here the
    yardstick is the one of tests/test_horizon_gpu.py: max(contract, diameter of the legal runs), factor 2.  Measured: 4096^2 1.8e-3 / 1.5e-4, 2048^2 5e-4 / 3e-4."""
    rs = _spread()
    key = f"bench_{size}_float_400x2"
    ref = _golden()[f"image_warping_{size}x{size}_float_gaussNewtonGPU_400"]["costs"]
    assert rs.n_reference_order_runs(key) >= 3
    P = wl.image_warping(size, size)
    g = hip_solver(P, nIterations=len(ref) - 1, lIterations=400)
    dev = api.to_device(P)
    g.init(dev)
    costs = [g.cost()]
    while g.step(dev):
        costs.append(g.cost())
    g.close()
    assert len(costs) == len(ref)
    assert abs(costs[0] - ref[0]) <= 1e-5 * abs(ref[0])
    for i in range(1, len(ref)):
        v = rs.verdict(key, "float", costs[i], i)
        assert v["within_reference_spread"], (i, costs, ref, v)


def test_benchmark_workload_2048_double_against_frozen_oracle_costs():
    """The same in double: rounding differences start at 1e-16 and are amplified by 400 PCG iterations to ~1e-4 in the cost -- the long-horizon trajectory is a property
    of the arithmetic order, not of the algorithm.  Yardstick as above, from the frozen double runs (exact-order plain / fma build, reference-order seeds)."""
    rs = _spread()
    key = "bench_2048_double_400x2"
    ref = _golden()["image_warping_2048x2048_double_gaussNewtonGPU_400"]["costs"]
    assert rs.n_reference_order_runs(key) >= 2
    P = wl.image_warping(2048, 2048, double=True)
    g = hip_solver(P, nIterations=len(ref) - 1, lIterations=400)
    dev = api.to_device(P)
    g.init(dev)
    costs = [g.cost()]
    while g.step(dev):
        costs.append(g.cost())
    g.close()
    assert abs(costs[0] - ref[0]) <= 1e-12 * ref[0]
    for i in range(1, len(ref)):
        v = rs.verdict(key, "double", costs[i], i)
        assert v["within_reference_spread"], (i, costs, ref, v)


# ---- (e) a fast-converging solve: the expanded beta numerator under cancellation -------------------------------------------------------
def test_fast_converging_solve_expanded_beta(oracle_lib, monkeypatch):
    """Every pixel constrained with a heavy fit weight: the system is nearly diagonal, the residual drops by orders of magnitude per
    iteration and betaNumerator = alphaNum - 2 alpha s2 + alpha^2 s3 cancels almost completely (clamped at 0 like the direct sum it replaces).
    The solve must still follow the oracle and the three-kernel loop."""
    P = wl.image_warping(300, 210, random_state=3, mask_fraction=0.02, perturb=0.3)
    P.params[3][...] = P.params[2] + 0.25                       # Constraints everywhere
    P.params[5] = np.array(np.sqrt(np.float32(1e6)), dtype=np.float32)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_ONEKERNEL", mode)
        g = hip_solver(P, nIterations=2, lIterations=30)
        dev = api.to_device(P)
        g.init(dev); c = [g.cost()]
        while g.step(dev):
            c.append(g.cost())
        res[mode] = (c, device_unknowns(P, dev)); g.close()
    o = oracle_solver(oracle_lib, P, nIterations=2, lIterations=30)
    Pref = P.clone(); o.solve(Pref.params)
    # the solve converges to float rounding (cost 1e10 -> 1e-4 -> 1e-15): below 1e-12 of the initial cost only the order of magnitude is meaningful
    atol = 1e-12 * res["1"][0][0]
    np.testing.assert_allclose(res["1"][0], o.cost_history(), rtol=1e-5, atol=atol)
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=1e-5, atol=atol)
    assert np.all(np.isfinite(res["1"][1])) and rel_err(res["1"][1], flat_unknowns(Pref)) < 1e-5
    o.close()


# ---- (f) one full-size step of configs 3 and 4 against the oracle ----------------------------------------------------------------------
def test_config3_sfs_1024_double_lm_step_vs_oracle(oracle_lib):
    P = wl.shape_from_shading(1024, 1024, double=True, holes=True)
    _pair(oracle_lib, P, "LMGPU", 1, 10, 1e-10, 1e-9, radius_tol=1e-8)


def test_config4_arap_500k_step_vs_oracle(oracle_lib):
    P = wl.arap_mesh_deformation(708, 707, perturb=0.01)
    _pair(oracle_lib, P, "gaussNewtonGPU", 1, 10, 1e-5, 1e-5)


def test_config4_arap_500k_lm_steps_vs_oracle(oracle_lib):
    """The same mesh under Levenberg-Marquardt (the reference's performance run of this example is GN and LM, arap_mesh_deformation/src/main.cpp:81-99): two outer steps of 12
    iterations on the two-kernel iteration of round 6 -- CtC in the plane gather, b / Q / delta-out in the flat pass, one split residual reset (period 10) with its restart
    launch -- at full size (500 556 vertices: 768 workgroups walking their XCD eighths)."""
    P = wl.arap_mesh_deformation(708, 707, perturb=0.01)
    _pair(oracle_lib, P, "LMGPU", 2, 12, 1e-5, 1e-5, radius_tol=1e-3)


def test_delta_placement_trial_changes_no_bit(monkeypatch, capfd):
    """Round 6: delta is the one vector the Gauss-Newton loop reads and writes, and the time of a launch follows the region the allocator put it in (profiles/NOTES.md).  The
    first long linear solve of a large single-GPU plan copies delta into a fresh vector every six launches (four candidates, each window timed) and goes on in the fastest
    (PcgSolver::deltaTrial).  A copy is a copy: unknowns and costs are the bits of a run with the trial switched off; the trial does run (its report names four timings) and a
    second solve on the same plan does not repeat it; a solve too short for four windows leaves everything as it is."""
    P = wl.image_warping(2400, 2400, random_state=6, perturb=0.3)
    res = []
    for trial in ("0", "2"):
        monkeypatch.setenv("OPT_AMD_DELTA_TRIAL", trial)
        g = hip_solver(P, "gaussNewtonGPU", nIterations=3, lIterations=40)
        dev = api.to_device(P)
        capfd.readouterr()
        g.init(dev)
        costs = [g.cost()]
        while g.step(dev):
            costs.append(g.cost())
        costs.append(g.cost())
        err = capfd.readouterr().err
        res.append((costs, device_unknowns(P, dev), err))
        assert g.describe().get("delta_placement_trial") == ("off" if trial == "0" else "done"), g.describe()      # (OptAmd_PlanDescribe says where the trial stands)
        g.close()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    assert "delta placement trial" not in res[0][2]
    assert res[1][2].count("delta placement trial") == 1 and len(res[1][2].split("launches:")[1].split("->")[0].split()) == 4, res[1][2]
    monkeypatch.setenv("OPT_AMD_DELTA_TRIAL", "2")
    g = hip_solver(P, "gaussNewtonGPU", nIterations=2, lIterations=20)      # 20 < 4 + 4 x 6 launches: no trial
    dev = api.to_device(P)
    capfd.readouterr()
    g.solve(dev)
    assert "delta placement trial" not in capfd.readouterr().err
    g.close()


def test_delta_placement_trial_on_the_march_template_changes_no_bit(monkeypatch, capfd):
    """The same trial on the marching template's Gauss-Newton loop (poisson_image_editing 2048^2: 64 MiB vectors, the smallest the trial looks at)."""
    P = wl.poisson_image_editing(2048, 2048, seed=3)
    res = []
    for trial in ("0", "2"):
        monkeypatch.setenv("OPT_AMD_DELTA_TRIAL", trial)
        g = hip_solver(P, "gaussNewtonGPU", nIterations=2, lIterations=40)
        dev = api.to_device(P)
        capfd.readouterr()
        g.solve(dev)
        res.append((g.cost(), device_unknowns(P, dev), capfd.readouterr().err))
        g.close()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    assert "delta placement trial" not in res[0][2] and res[1][2].count("delta placement trial") == 1, res[1][2]
