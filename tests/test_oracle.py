"""CPU tests of the oracle itself (-m "not gpu").

The reference ships no golden vectors, so the oracle is pinned by
  * the reference's one known-answer test, tests/minimal_graph_only (curve fit -> (100,102)),
  * identities any correct restatement of Opt's generator must satisfy: J^T F equals the finite-difference
    gradient of the cost, J^T J is symmetric PSD with diag equal to the preconditioner diagonal, and
  * the committed golden fixtures (tests/golden), which freeze today's oracle output against regressions.
"""
import numpy as np
import pytest

from opt_amd import workloads as wl
from helpers import flat_unknowns, oracle_solver, rel_err


def test_known_answer_curve_fitting(oracle_lib):
    # reference tests/minimal_graph_only/main.cpp:43-61, 88-90: double precision, GN defaults (10 x 10)
    P = wl.curve_fitting()
    s = oracle_solver(oracle_lib, P)
    s.solve(P.params)
    a, b = P.params[0].reshape(-1)
    assert abs(a - 100.0) < 1e-9 and abs(b - 102.0) < 1e-9
    assert s.cost() < 1e-12
    hist = s.cost_history()
    assert hist[0] > 1e6 and np.all(np.diff(hist[:6]) < 0)


def _fd_gradient(s, P, h):
    g = []
    for slot in P.unknown_slots:
        X = P.params[slot].reshape(-1)
        for i in range(X.size):
            old = X[i]
            X[i] = old + h; cp = s.eval_cost(P.params)
            X[i] = old - h; cm = s.eval_cost(P.params)
            X[i] = old
            g.append((cp - cm) / (2 * h))
    return np.array(g)


def _active_mask(P):
    if P.energy == "image_warping":
        m = np.asarray(P.params[4]).reshape(-1) == 0
        return np.concatenate([np.repeat(m, 2), m])
    if P.energy == "poisson_image_editing":
        return np.repeat(np.asarray(P.params[2]).reshape(-1) == 0, 4)
    if P.energy == "shape_from_shading":
        return np.asarray(P.params[17]).reshape(-1) > 0
    return np.ones(flat_unknowns(P).size, dtype=bool)


CASES = {
    "image_warping": lambda: wl.image_warping(9, 7, double=True, random_state=1, mask_fraction=0.1, perturb=0.5),
    "poisson_image_editing": lambda: wl.poisson_image_editing(8, 6, double=True, seed=3),
    "arap_mesh_deformation": lambda: wl.arap_mesh_deformation(5, 4, double=True, seed=2, perturb=0.02),
    "curveFitting": lambda: wl.curve_fitting(16, double=True),
    "shape_from_shading": lambda: wl.shape_from_shading(12, 10, double=True, seed=4, holes=True, noise=2e-3),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_jtf_is_gradient_of_cost(oracle_lib, name):
    """F^ = J^T F must be d(cost)/dx on non-excluded rows.  For energies whose excluded centres carry
    residuals that touch active unknowns (poisson), the cost drops those rows while J^T F keeps them
    (o.t:2045-2064 has no exclude test) -- so the identity is checked with no pixel excluded there."""
    P = CASES[name]()
    if name == "poisson_image_editing":
        P.params[2][...] = 0
    s = oracle_solver(oracle_lib, P)
    f, d = s.eval_jtf(P.params)
    g = _fd_gradient(s, P, 1e-7 if name == "shape_from_shading" else 1e-6)
    act = _active_mask(P)
    assert rel_err(f[act], g[act]) < (1e-5 if name == "shape_from_shading" else 1e-6)
    assert np.all(d >= 0)


@pytest.mark.parametrize("name", sorted(CASES))
def test_jtj_symmetric_psd_with_matching_diagonal(oracle_lib, name):
    P = CASES[name]()
    s = oracle_solver(oracle_lib, P)
    n = s.n
    act = _active_mask(P)
    rng = np.random.default_rng(0)
    a = rng.standard_normal(n) * act
    b = rng.standard_normal(n) * act
    Aa, Ab = s.apply_jtj(P.params, a), s.apply_jtj(P.params, b)
    assert abs(a @ Ab - b @ Aa) <= 1e-10 * (abs(a @ Ab) + 1)
    assert a @ Aa >= 0
    _, d = s.eval_jtf(P.params)
    idx = np.flatnonzero(act)[:: max(1, act.sum() // 25)]
    for i in idx:
        e = np.zeros(n); e[i] = 1
        assert abs(s.apply_jtj(P.params, e)[i] - d[i]) <= 1e-10 * (abs(d[i]) + 1)


def test_poisson_excluded_neighbours_still_contribute(oracle_lib):
    """SURVEY 8a 'exclude' row: residuals centred on excluded pixels still feed their active neighbours'
    J^T F / J^T J, so each in-bounds edge counts twice in diag(J^T J) regardless of the mask."""
    P = wl.poisson_image_editing(8, 8, double=True)
    s = oracle_solver(oracle_lib, P)
    _, d = s.eval_jtf(P.params)
    d = d.reshape(8, 8, 4)
    assert np.all(d[3, 3] == 8.0) and np.all(d[0, 0] == 4.0) and np.all(d[0, 3] == 6.0)


def test_gn_reduces_cost_image_warping(oracle_lib):
    P = wl.image_warping(24, 20, double=False)
    s = oracle_solver(oracle_lib, P, nIterations=3, lIterations=20)
    s.solve(P.params)
    h = s.cost_history()
    assert len(h) == 4 and h[-1] < h[0] * 0.5
    tr = s.trace()
    assert tr.shape == (60, 6) and np.all(tr[:, 3] > 0)


def test_lm_accepts_and_updates_radius(oracle_lib):
    P = wl.image_warping(16, 16, double=True)
    s = oracle_solver(oracle_lib, P, "LMGPU", nIterations=4, lIterations=30)
    s.init(P.params)
    c0 = s.cost()
    while s.step(P.params):
        pass
    assert s.cost() < c0
    assert s.trust_region_radius() != 1e4


def test_unknown_energy_or_kind_rejected(oracle_lib):
    with pytest.raises(ValueError):
        oracle_lib.OracleSolver("no_such_energy", "gaussNewtonGPU", False, (4, 4))
    with pytest.raises(ValueError):
        oracle_lib.OracleSolver("image_warping", "gradientDescentCPU", False, (4, 4))   # o.t:122
