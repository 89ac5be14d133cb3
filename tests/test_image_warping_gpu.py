"""GPU parity tests for image_warping (-m gpu): every stage of the HIP path against the CPU oracle on the
same seeded inputs, through the C ABI (Opt.h + the OptAmd.h probes).

Tolerances: float results must agree to 1e-5 relative (BASELINE.json north_star), double to 1e-12 on
costs; per-vector checks use the same bar relative to the vector norm.
"""
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

SIZES = [(64, 64), (257, 129), (31, 5), (300, 1), (1, 40)]


def _problem(W, H, double, seed=7, jitter=0.0):
    # jitter = 0: rest shape is the pixel lattice (the reference example's case, lattice fast path);
    # jitter > 0: arbitrary rest shape (general kernels)
    return wl.image_warping(W, H, double=double, random_state=seed, mask_fraction=0.08, perturb=0.4, jitter_urshape=jitter)


@pytest.mark.parametrize("jitter", [0.0, 0.2])
@pytest.mark.parametrize("W,H", SIZES)
@pytest.mark.parametrize("double", [False, True])
def test_cost_jtf_diag_jtjp(oracle_lib, W, H, double, jitter):
    import torch
    P = _problem(W, H, double, jitter=jitter)
    tol = 1e-11 if double else 2e-5
    o = oracle_solver(oracle_lib, P)
    g = hip_solver(P)
    dev = api.to_device(P)
    # cost
    c_ref, c_gpu = o.eval_cost(P.params), g.eval_cost(dev)
    assert abs(c_gpu - c_ref) <= (1e-12 if double else 1e-5) * abs(c_ref)
    # J^T F and diag(J^T J)
    f_ref, d_ref = o.eval_jtf(P.params)
    f_gpu, d_gpu = g.eval_jtf(dev)
    act = np.concatenate([np.repeat(P.params[4].reshape(-1) == 0, 2), P.params[4].reshape(-1) == 0])
    assert rel_err(f_gpu.cpu().numpy()[act], f_ref[act]) < tol
    assert rel_err(d_gpu.cpu().numpy()[act], d_ref[act]) < tol
    assert np.all(f_gpu.cpu().numpy()[~act] == 0)          # excluded rows are written as zero
    # J^T J p for a seeded p (zero on excluded rows, as in the solver)
    rng = np.random.default_rng(11)
    v = (rng.standard_normal(o.n) * act).astype(o.dtype)
    Av_ref = o.apply_jtj(P.params, v)
    Av_gpu, dot = g.apply_jtj(dev, torch.from_numpy(v).cuda())
    assert rel_err(Av_gpu.cpu().numpy(), Av_ref) < tol
    assert abs(dot - float(v.astype(np.float64) @ Av_ref.astype(np.float64))) <= 10 * tol * abs(dot)
    g.close(); o.close()


@pytest.mark.parametrize("jitter", [0.0, 0.2])
@pytest.mark.parametrize("double", [False, True])
def test_gn_trajectory(oracle_lib, double, jitter):
    """3 GN x 10 PCG on 48x40: per-PCG-iteration scalars, per-GN cost and final unknowns."""
    P = _problem(48, 40, double, seed=3, jitter=jitter)
    o = oracle_solver(oracle_lib, P, nIterations=3, lIterations=10)
    g = hip_solver(P, nIterations=3, lIterations=10)
    g.enable_trace()
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    costs_o, costs_g = [o.cost()], [g.cost()]
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b
        if not a:
            break
        costs_o.append(o.cost()); costs_g.append(g.cost())
    tol = 1e-11 if double else 1e-5
    np.testing.assert_allclose(costs_g, costs_o, rtol=tol)
    to, tg = o.trace(), g.trace()
    assert to.shape == tg.shape == (30, 6)
    np.testing.assert_allclose(tg[:, 2:5], to[:, 2:5], rtol=1e-9 if double else 5e-3)   # float PCG scalars amplify last-bit differences; the contract is the cost
    assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, 1e-11 if double else 1e-5, absolute=True, double=double)
    g.close(); o.close()


def test_lm_trajectory_double(oracle_lib):
    P = _problem(40, 36, True, seed=5)
    kw = dict(nIterations=5, lIterations=25)
    o = oracle_solver(oracle_lib, P, "LMGPU", **kw)
    g = hip_solver(P, "LMGPU", **kw)
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    assert_close("cost", g.cost(), o.cost(), 1e-12, double=True)
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b
        assert_close("cost", g.cost(), o.cost(), 1e-10, double=True)
        assert_close("radius", g.trust_region_radius(), o.trust_region_radius(), 1e-8, double=True)
        if not a:
            break
    assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, 1e-9, absolute=True, double=True)
    g.close(); o.close()


def test_lm_trajectory_float(oracle_lib):
    P = _problem(40, 36, False, seed=5)
    kw = dict(nIterations=4, lIterations=25)
    o = oracle_solver(oracle_lib, P, "LMGPU", **kw)
    g = hip_solver(P, "LMGPU", **kw)
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b
        assert abs(g.cost() - o.cost()) <= 1e-5 * o.cost()
        if not a:
            break
    g.close(); o.close()


def test_solve_matches_init_plus_steps_and_is_deterministic():
    import torch
    P = wl.image_warping(96, 80)
    outs = []
    for mode in ("solve", "steps", "solve"):
        g = hip_solver(P, nIterations=2, lIterations=15)
        dev = api.to_device(P)
        if mode == "solve":
            g.solve(dev)
        else:
            g.init(dev)
            while g.step(dev):
                pass
        outs.append((g.cost(), torch.cat([dev[0].reshape(-1), dev[1].reshape(-1)]).cpu().numpy()))
        g.close()
    assert outs[0][0] == outs[1][0] == outs[2][0]            # bitwise: reductions are order-fixed
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][1], outs[2][1])


def test_full_size_properties():
    """2048^2 (BASELINE config 2): size-independent properties instead of an oracle run --
    symmetry p.Aq = q.Ap, p.Ap >= 0, A(e_i)_i = diag_i, and GN monotonically reduces the cost."""
    import torch
    W = H = 2048
    P = wl.image_warping(W, H)
    g = hip_solver(P, nIterations=2, lIterations=10)
    dev = api.to_device(P)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    p = torch.randn(g.n, device="cuda", generator=gen); q = torch.randn(g.n, device="cuda", generator=gen)
    Ap, pAp = g.apply_jtj(dev, p)
    Aq, _ = g.apply_jtj(dev, q)
    pAq, qAp = float(p.double() @ Aq.double()), float(q.double() @ Ap.double())
    assert abs(pAq - qAp) <= 1e-4 * max(abs(pAq), abs(qAp), 1.0)
    assert pAp > 0 and abs(pAp - float(p.double() @ Ap.double())) <= 1e-5 * pAp
    _, diag = g.eval_jtf(dev)
    for i in (0, 12345, 2 * W * H + 777, g.n - 1):
        e = torch.zeros(g.n, device="cuda"); e[i] = 1
        Ae, _ = g.apply_jtj(dev, e)
        assert abs(float(Ae[i]) - float(diag[i])) <= 1e-5 * abs(float(diag[i]))
    g.init(dev); c = [g.cost()]
    while g.step(dev):
        c.append(g.cost())
    assert len(c) == 3 and c[2] < c[1] < c[0]
    g.close()


@pytest.mark.parametrize("double", [False, True])
def test_single_kernel_iteration_matches_three_kernel_loop(double, monkeypatch):
    """The fused one-kernel-per-iteration PCG loop evaluates the beta numerator by expanding sum M (r - alpha Ap)^2
    (energy.h PcgIterArgs).  Over a long solve (2 GN x 200 PCG) it must track the Step1/Step2/Step3 loop: identical
    math, so double agrees to 1e-9 and float to the 1e-5 cost bar."""
    P = wl.image_warping(200, 160, double=double, random_state=21, mask_fraction=0.05, perturb=0.4)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_ONEKERNEL", mode)
        g = hip_solver(P, nIterations=2, lIterations=200)
        dev = api.to_device(P)
        g.init(dev); costs = [g.cost()]
        while g.step(dev):
            costs.append(g.cost())
        res[mode] = (costs, device_unknowns(P, dev))
        g.close()
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=1e-9 if double else 1e-5)
    assert rel_err(res["1"][1], res["0"][1]) < (1e-8 if double else 1e-4)


def test_lattice_fast_path_matches_the_general_path(monkeypatch):
    """The unit-lattice kernel drops the UrShape loads; its arithmetic is the general kernel's with U_c - U_n = -n folded
    in (only FMA contraction may differ), so on a lattice input both must agree to rounding."""
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_LATTICE", flag)
        P = wl.image_warping(150, 70, random_state=4, mask_fraction=0.05, perturb=0.3)
        g = hip_solver(P, nIterations=2, lIterations=25)
        dev = api.to_device(P)
        g.solve(dev)
        outs.append((g.cost(), device_unknowns(P, dev)))
        g.close()
    assert abs(outs[0][0] - outs[1][0]) <= 2e-6 * outs[1][0] and rel_err(outs[0][1], outs[1][1]) < 2e-6


RAGGED = [(1, 1), (2, 1), (1, 6), (3, 3), (59, 2), (60, 5), (61, 3), (63, 4), (64, 2), (65, 7), (119, 3), (121, 6), (719, 2), (720, 3), (721, 5), (1441, 4), (7, 301)]


@pytest.mark.parametrize("liters", [1, 2, 3, 7, 8])
@pytest.mark.parametrize("W,H", RAGGED)
def test_iteration_kernel_on_ragged_sizes(oracle_lib, W, H, liters):
    """The single-kernel PCG iteration tiles an image into 60-pixel wave spans (720-pixel strips), keeps a 2-pixel ring and marches
    three rows per pass: sizes around every one of those boundaries, down to 1x1, for odd and even launch counts (the delta
    update is paired over two launches), against the oracle in double."""
    P = wl.image_warping(W, H, double=True, random_state=W * 31 + H, mask_fraction=0.1 if W * H > 8 else 0.0, perturb=0.3)
    o = oracle_solver(oracle_lib, P, nIterations=2, lIterations=liters)
    g = hip_solver(P, nIterations=2, lIterations=liters)
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    scale = max(abs(o.cost()), 1e-300)
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b
        assert abs(g.cost() - o.cost()) <= 1e-10 * max(abs(o.cost()), 1e-12 * scale)      # a 1x1 image converges to cost ~1e-31
        if not a:
            break
    assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, 1e-9, absolute=True, double=True)
    g.close(); o.close()


SWITCHES = [{"OPT_AMD_ONEKERNEL": "0"}, {"OPT_AMD_LATTICE": "0"}, {"OPT_AMD_ITER_ROWS": "5"}]


@pytest.mark.parametrize("jitter", [0.0, 0.04])
@pytest.mark.parametrize("liters", [1, 2, 3, 6, 9])
def test_switches_keep_the_result(monkeypatch, jitter, liters):
    """INTEGRATION.md section 6 lists the environment switches that are left: the reference-ordered three-kernel loop (the parity control; with and without
    PCGStep3 fused into the next PCGStep1), the general-UrShape kernels on a lattice input, forced rows per workgroup.  Every one of them must solve the same
    problem: two Gauss-Newton steps in double against the default path, on a unit lattice and on a general UrShape, for launch counts on either side of the paired
    delta update and of the first launch that writes delta."""
    def run(env):
        for k in {k for sw in SWITCHES for k in sw}:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        P = wl.image_warping(233, 140, double=True, random_state=9, mask_fraction=0.06, perturb=0.3, jitter_urshape=jitter)
        g = hip_solver(P, nIterations=2, lIterations=liters)
        dev = api.to_device(P)
        g.init(dev); costs = [g.cost()]
        while g.step(dev):
            costs.append(g.cost())
        out = (costs, device_unknowns(P, dev)); g.close()
        return out
    ref = run({})
    for sw in SWITCHES:
        got = run(sw)
        np.testing.assert_allclose(got[0], ref[0], rtol=1e-9, err_msg=str(sw))
        assert rel_err(got[1], ref[1]) < 1e-8, sw


@pytest.mark.parametrize("double,liters", [(True, 1), (True, 2), (True, 3), (True, 9), (False, 8)])
def test_half_lattice_image_takes_the_general_kernel(oracle_lib, double, liters):
    """UrShape is the unit lattice on the upper half of the image and jittered below: the lattice verdict is per image, so the general kernel runs everywhere -- with
    M_a rebuilt per pixel from the pairs its stencil evaluates (no preconditioner stream), on lattice and off-lattice rows alike, masks and sweep directions included."""
    P = wl.image_warping(197, 120, double=double, random_state=31, mask_fraction=0.07, perturb=0.3)
    rng = np.random.default_rng(5)
    P.params[2][60:] += (0.2 * rng.standard_normal(P.params[2][60:].shape)).astype(P.params[2].dtype)
    o = oracle_solver(oracle_lib, P, nIterations=2, lIterations=liters)
    g = hip_solver(P, nIterations=2, lIterations=liters, timing=True)
    Pref = P.clone(); dev = api.to_device(P)
    o.init(Pref.params); g.init(dev)
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b
        assert_close("cost", g.cost(), o.cost(), 1e-10 if double else 1e-5, double=double)
        if not a:
            break
    assert "PCGIteration" in g.kernel_timings()
    assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, 1e-9 if double else 2e-5, absolute=True, double=double)
    g.close(); o.close()


def test_urshape_leaves_the_lattice_between_two_steps(oracle_lib):
    """The marching PCGInit1 of step n runs on the lattice verdict of step n - 1's bind while its own bind's verdict is in flight; a caller that moves UrShape off the
    unit lattice between two Opt_ProblemStep calls (in place, same buffers) makes that guess wrong: the lattice variant has to be redone as the general one before the first
    PCG launch (ImageWarpingOps::pcgIteration).  Both steps against the oracle, which is given the same edit; then back onto the lattice for a third step."""
    import torch
    P = wl.image_warping(150, 97, double=True, random_state=21, mask_fraction=0.05, perturb=0.3)
    o = oracle_solver(oracle_lib, P, nIterations=3, lIterations=9)
    g = hip_solver(P, nIterations=3, lIterations=9)
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    assert o.step(Pref.params) and g.step(dev)
    assert_close("cost", g.cost(), o.cost(), 1e-10, double=True)
    rng = np.random.default_rng(5)
    lattice = np.array(Pref.params[2], copy=True)
    jit = 0.03 * rng.standard_normal(Pref.params[2].shape)
    Pref.params[2][...] = lattice + jit                               # UrShape (binding index 2), in place on both sides
    dev[2].copy_(torch.from_numpy(lattice + jit).cuda())
    assert o.step(Pref.params) and g.step(dev)
    assert_close("cost", g.cost(), o.cost(), 1e-10, double=True)
    Pref.params[2][...] = lattice
    dev[2].copy_(torch.from_numpy(lattice).cuda())
    o.step(Pref.params); g.step(dev)
    assert_close("cost", g.cost(), o.cost(), 1e-10, double=True)
    assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, 1e-9, absolute=True, double=True)
    g.close(); o.close()


@pytest.mark.parametrize("double", [False, True])
def test_real_cat_mask(oracle_lib, double):
    """image_warping on the reference example's own mask (examples/data/cat512_mask.png sampled every 4th pixel,
    tests/fixtures/cat_mask_128.npz) with the cat512 markers: an irregular solved region with holes and one-pixel-wide parts.
    Every stage (cost, J^T F, diag, J^T J p) must match the oracle.  With nine point constraints the system is close to singular
    and PCG amplifies summation-order differences ~30x per iteration (2 iterations: 1e-17 relative in the unknowns, 5: 1e-13,
    40: 1e-4 -- the three-kernel loop, the single-kernel loop and the oracle drift apart at that rate from bit-identical stage
    outputs), so the solve itself is compared in double over one Gauss-Newton step of 5 PCG iterations."""
    import os
    import torch
    from opt_amd import io
    m = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "cat_mask_128.npz"))["mask_red"]
    P = io.image_warping_problem_from_mask(m, downsample=1, double=double)
    for sx, sy, tx, ty in wl.CAT512_MARKERS:
        P.params[3][sy // 4, sx // 4] = (tx / 4.0, ty / 4.0)
    tol = 1e-11 if double else 2e-5
    o = oracle_solver(oracle_lib, P, nIterations=1, lIterations=5)
    g = hip_solver(P, nIterations=1, lIterations=5)
    dev = api.to_device(P)
    assert abs(g.eval_cost(dev) - o.eval_cost(P.params)) <= (1e-12 if double else 1e-5) * abs(o.eval_cost(P.params))
    f_ref, d_ref = o.eval_jtf(P.params); f_gpu, d_gpu = g.eval_jtf(dev)
    act = np.concatenate([np.repeat(P.params[4].reshape(-1) == 0, 2), P.params[4].reshape(-1) == 0])
    assert rel_err(f_gpu.cpu().numpy()[act], f_ref[act]) < tol and rel_err(d_gpu.cpu().numpy()[act], d_ref[act]) < tol
    v = (np.random.default_rng(5).standard_normal(o.n) * act).astype(o.dtype)
    Av_gpu, _ = g.apply_jtj(dev, torch.from_numpy(v).cuda())
    assert rel_err(Av_gpu.cpu().numpy(), o.apply_jtj(P.params, v)) < tol
    if double:
        Pref = P.clone()
        o.init(Pref.params); g.init(dev)
        o.step(Pref.params); g.step(dev)
        assert_close("cost", g.cost(), o.cost(), 1e-9, double=True)
        assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, 1e-9, absolute=True, double=True)
    g.close(); o.close()


@pytest.mark.parametrize("double", [False, True])
def test_lm_single_kernel_loop_matches_three_kernel_loop(double, monkeypatch):
    """Levenberg-Marquardt on the A p-free single-kernel iteration (Q delivered one launch late, split residual reset every 10th
    iteration followed by a restart launch, q early-out) against the Step1 / Step2 / Step3 loop: costs, trust-region radius, unknowns."""
    P = wl.image_warping(150, 131, double=double, random_state=23, mask_fraction=0.05, perturb=0.4)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("OPT_AMD_ONEKERNEL_LM", mode)
        g = hip_solver(P, "LMGPU", nIterations=4, lIterations=45)
        dev = api.to_device(P)
        g.init(dev); costs = [g.cost()]; radii = []
        while g.step(dev):
            costs.append(g.cost()); radii.append(g.trust_region_radius())
        res[mode] = (costs, radii, device_unknowns(P, dev))
        g.close()
    assert len(res["1"][0]) == len(res["0"][0])
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=1e-9 if double else 1e-5)
    np.testing.assert_allclose(res["1"][1], res["0"][1], rtol=1e-7 if double else 1e-3)
    assert rel_err(res["1"][2], res["0"][2]) < (1e-8 if double else 1e-4)


def test_timing_can_be_switched_on_between_steps():
    """OptAmd_PlanSetTiming: per-kernel hipEvents from the next launch on, totals restart; the solve itself is unaffected (bench.py uses it for the
    roofline leg of a multi-GPU job: timed steps without events, then two steps with them on the same plan)."""
    P = wl.image_warping(96, 64, random_state=2, perturb=0.3)
    ref = hip_solver(P, "gaussNewtonGPU", nIterations=4, lIterations=6)
    dev_ref = api.to_device(P)
    ref.init(dev_ref)
    want = []
    while ref.step(dev_ref):
        want.append(ref.cost())
    ref.close()

    g = hip_solver(P, "gaussNewtonGPU", nIterations=4, lIterations=6)
    dev = api.to_device(P)
    g.init(dev)
    assert g.step(dev)
    assert g.kernel_timings().get("PCGIteration", (0, 0.0))[0] == 0        # nothing was timed
    got = [g.cost()]
    g.set_timing(True)
    assert g.step(dev)
    got.append(g.cost())
    kt = g.kernel_timings()
    assert kt["PCGIteration"][0] == 6 and kt["PCGIteration"][1] > 0           # exactly the launches of this step
    g.set_timing(False)
    assert g.step(dev)
    got.append(g.cost())
    assert g.kernel_timings().get("PCGIteration", (0, 0.0))[0] == 0        # totals restart when the switch is flipped
    g.close()
    assert got == want[:3]
