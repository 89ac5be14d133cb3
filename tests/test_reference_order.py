"""CPU tests (no GPU) of the oracle's reference-order reduction mode and of the yardstick built on it (tools/reference_spread.py).

oracle/solver.hpp reductionMode 1 restates how the REFERENCE adds up a dot product: one opt_float term per element of the index space, a 32-lane shfl.down tree
over the 16 x 2 pixel patch of a warp (API/src/util.t:612-623), one opt_float atomicAdd per warp (API/src/solverGPUGaussNewton.t:312-317) -- committed in a seeded
random order, since the hardware does not define one.  Every seed is one legal run of the reference; their spread is the yardstick of the long-horizon parity
tests (tests/test_horizon_gpu.py).  Here: the mode is deterministic per seed and independent of the thread count, seeds differ from each other and from the
exact-order sums by float rounding only, double runs agree to double rounding, and the frozen runs under tests/golden/ are complete and consistent.
"""
import os
import sys

import numpy as np
import pytest

from opt_amd import workloads as wl
from helpers import flat_unknowns, oracle_solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import reference_spread as rs      # noqa: E402


def _run(oracle_lib, double, mode, seed, threads=1, liters=12, steps=2, size=(96, 70)):
    P = wl.image_warping(*size, double=double, random_state=3, mask_fraction=0.05, perturb=0.3)
    o = oracle_solver(oracle_lib, P, "gaussNewtonGPU", nIterations=steps, lIterations=liters)
    o.set_threads(threads)
    o.set_reduction(mode, seed)
    o.init(P.params)
    costs = [o.cost()]
    while o.step(P.params):
        costs.append(o.cost())
    costs.append(o.cost())
    x = flat_unknowns(P)
    o.close()
    return np.array(costs), x


def test_same_seed_same_bits(oracle_lib):
    """A seed fixes the run: repeated, and whatever the number of threads of the banded traversal (the single-threaded oracle scatters J^T J p in raster order, the
    banded one in two colours of row bands -- another legal order of the same float additions, so threads = 1 is a different run)."""
    a, xa = _run(oracle_lib, False, 1, 7, threads=1)
    b, xb = _run(oracle_lib, False, 1, 7, threads=1)
    assert np.array_equal(a, b) and np.array_equal(xa, xb)
    c, xc = _run(oracle_lib, False, 1, 7, threads=2)
    d, xd = _run(oracle_lib, False, 1, 7, threads=4)
    assert np.array_equal(c, d) and np.array_equal(xc, xd)
    assert np.all(np.abs(a - c) <= 1e-5 * np.abs(a))


def test_seeds_are_different_legal_runs_float(oracle_lib):
    exact, _ = _run(oracle_lib, False, 0, 0)
    runs = [_run(oracle_lib, False, 1, s)[0] for s in (1, 2, 3, 4)]
    assert len({tuple(r) for r in runs}) > 1                      # the commit order matters in float
    for r in runs:
        assert r[0] == pytest.approx(exact[0], rel=1e-6)          # the initial cost: one sum, float rounding only
        assert np.all(np.abs(r - exact) <= 1e-3 * np.abs(exact))  # a short solve: still close to the exact-order run
        assert np.all(np.diff(r[:2]) < 0)                         # and it does descend


def test_double_runs_agree_to_double_rounding(oracle_lib):
    exact, xe = _run(oracle_lib, True, 0, 0)
    for s in (1, 2):
        r, x = _run(oracle_lib, True, 1, s)
        assert np.all(np.abs(r - exact) <= 1e-10 * np.abs(exact))
        assert np.max(np.abs(x - xe)) <= 1e-9 * np.max(np.abs(xe))


def test_graph_energies_scatter_in_a_seeded_edge_order(oracle_lib):
    """Graph energies: the reference scatters J^T F / J^T J p with one float atomic per (vertex, channel) of every hyperedge, in no defined order; mode 1 visits the
    hyperedges in a seeded random permutation (global sums stay exact).  Same seed, same bits; different seeds, float-rounding apart; double, double-rounding apart."""
    def run(double, mode, seed):
        P = wl.arap_mesh_deformation(23, 17, double=double, seed=3, perturb=0.01)
        o = oracle_solver(oracle_lib, P, "gaussNewtonGPU", nIterations=3, lIterations=12)
        o.set_reduction(mode, seed)
        o.solve(P.params)
        c = o.cost(); o.close()
        return c, flat_unknowns(P)
    a, b, c, e = run(False, 1, 5), run(False, 1, 5), run(False, 1, 6), run(False, 0, 0)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    assert not np.array_equal(a[1], c[1])
    for r in (a, c):
        assert abs(r[0] - e[0]) <= 1e-4 * abs(e[0])
    d0, d1 = run(True, 0, 0), run(True, 1, 5)
    assert abs(d0[0] - d1[0]) <= 1e-10 * abs(d0[0])


# ---- the frozen runs and the yardstick -------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("family,size", [("horizon", 2048), ("adversarial", 1024)])
@pytest.mark.parametrize("precision,nplain", [("float", 5), ("double", 3)])
@pytest.mark.parametrize("L", [20, 50, 100, 200, 400])
def test_frozen_runs_complete(family, size, precision, nplain, L):
    key = f"{family}_{size}_{precision}_{L}"
    runs = rs.legal_runs(key)
    labels = [l for l, _ in runs]
    assert "exact-order plain" in labels and "exact-order fma" in labels
    assert sum(l.startswith("reference-order plain") for l in labels) >= nplain
    if precision == "float":
        assert sum(l.startswith("reference-order fma") for l in labels) >= 3
    a = rs.anchor(key)
    for _, c in runs:
        assert abs(c - a) <= 0.05 * abs(a)          # legal runs, not different problems
    y = rs.yardstick(key, precision)
    assert y >= rs.FLOOR[precision] and y == max(rs.FLOOR[precision], rs.spread(key))
    assert rs.seed_spread(key) <= rs.spread(key)


def test_the_reference_does_not_meet_its_contract_against_itself_at_long_horizons():
    """What the yardstick is for: after 100+ float iterations two runs of the reference's own arithmetic are 1e-3 apart (contract: 1e-5)."""
    for L in (100, 200, 400):
        assert rs.seed_spread(f"horizon_2048_float_{L}") > 1e-4
    assert rs.seed_spread("horizon_2048_float_20") > 1e-5


def test_bench_keys_reuse_the_first_steps_of_the_frozen_solves():
    """bench_<size>_<precision>_400x2 = the first two Gauss-Newton steps of bench.py's workload: anchored on tests/golden/bench_costs.json, its legal runs include the
    first two steps of the frozen 8 x 400 solves of the same size (same workload, same initial guess), and at 4096^2 its own reference-order runs."""
    for key, prec, nref in (("bench_4096_float_400x2", "float", 5), ("bench_2048_float_400x2", "float", 3), ("bench_2048_double_400x2", "double", 3)):
        assert rs.n_reference_order_runs(key) >= nref, key
        for step in (1, 2):
            a = rs.anchor(key, step)
            assert a is not None and a == rs.anchor(key.replace("bench_", "solve8_").replace("_400x2", ""), step)      # the same exact-order oracle run, frozen twice
            assert rs.spread(key, step) > rs.FLOOR[prec]
            assert rs.verdict(key, prec, a, step)["within_reference_spread"]
    # what the yardstick says about the metric's own size: after the first 400 float iterations the reference's legal runs are ~1e-2 apart
    assert 1e-3 < rs.spread("bench_4096_float_400x2", 1) < 5e-2


def test_verdict_logic():
    key = "horizon_2048_float_400"
    a, y = rs.anchor(key), rs.yardstick(key, "float")
    assert rs.verdict(key, "float", a)["within_reference_spread"] and rs.verdict(key, "float", a)["within_contract"]
    v = rs.verdict(key, "float", a * (1 + 0.95 * rs.FACTOR * y))
    assert v["within_reference_spread"] and not v["within_contract"]
    assert not rs.verdict(key, "float", a * (1 + 1.05 * rs.FACTOR * y))["within_reference_spread"]
    assert rs.FACTOR == 1.0      # round 5: no allowance on top of the diameter of the legal runs (round 4: 2)
    assert rs.verdict("no_such_key", "float", 1.0) is None
