"""Golden-fixture tests.  tests/golden/*.npz freeze the CPU oracle's outputs on small seeded problems
(generator: tests/golden/generate.py; they are oracle outputs, not reference outputs -- the reference cannot run here).
CPU: the oracle still reproduces them.  GPU: the HIP path matches them through the C ABI."""
import glob
import os

import numpy as np
import pytest

from opt_amd import workloads as wl
from helpers import rel_err

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _load(path):
    z = np.load(path)
    n = sum(1 for k in z.files if k.startswith("param_"))
    params = [np.array(z[f"param_{i}"]) for i in range(n)]
    P = wl.Problem(str(z["energy"]), tuple(int(x) for x in z["dims"]), params, tuple(int(x) for x in z["unknown_slots"]), True)
    return z, P, str(z["kind"])


def test_fixtures_exist():
    assert len(GOLDEN) >= 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle_lib, path):
    z, P, kind = _load(path)
    o = oracle_lib.OracleSolver(P.energy, kind, True, P.dims)
    assert abs(o.eval_cost(P.params) - float(z["cost"])) <= 1e-13 * abs(float(z["cost"])) + 1e-300
    f, d = o.eval_jtf(P.params)
    assert rel_err(f, z["jtf"]) < 1e-13 and rel_err(d, z["diag"]) < 1e-13
    assert rel_err(o.apply_jtj(P.params, z["p"]), z["jtjp_unmasked"]) < 1e-13
    o.set("nIterations", int(z["n_iterations"]) if "n_iterations" in z.files else 3); o.set("lIterations", 10)
    o.solve(P.params)
    np.testing.assert_allclose(o.cost_history(), z["cost_history"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(o.trace(), z["trace"], rtol=1e-10, atol=1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_matches_golden(path):
    import torch
    from opt_amd import api
    from helpers import active_mask, device_unknowns, hip_solver
    z, P, kind = _load(path)
    g = hip_solver(P, kind, nIterations=int(z["n_iterations"]) if "n_iterations" in z.files else 3, lIterations=10)
    dev = api.to_device(P)
    assert abs(g.eval_cost(dev) - float(z["cost"])) <= 1e-12 * abs(float(z["cost"])) + 1e-300
    act = active_mask(P)
    f, d = g.eval_jtf(dev)
    assert rel_err(f.cpu().numpy()[act], z["jtf"][act]) < 1e-11 and rel_err(d.cpu().numpy()[act], z["diag"][act]) < 1e-11
    v = z["p"] * act
    Av, _ = g.apply_jtj(dev, torch.from_numpy(v).cuda())
    # golden J^T J p was taken with p non-zero everywhere; compare on a p masked like the solver's (excluded rows 0)
    from oracle.binding import OracleSolver
    o = OracleSolver(P.energy, kind, True, P.dims)
    assert rel_err(Av.cpu().numpy(), o.apply_jtj(P.params, v)) < 1e-11
    g.enable_trace()
    g.init(dev); costs = [g.cost()]
    while g.step(dev):
        costs.append(g.cost())
    hist = z["cost_history"]
    np.testing.assert_allclose(costs, hist[:len(costs)], rtol=1e-9, atol=1e-18)
    t = g.trace()
    # PCG scalars of a converged solve are round-off noise (curve fit reaches cost 0): compare relative to the first iteration's scale
    np.testing.assert_allclose(t[:, 2:5], z["trace"][:len(t), 2:5], rtol=1e-7, atol=1e-16 * np.abs(z["trace"][0, 2:5]).max())
    assert rel_err(device_unknowns(P, dev), z["final_unknowns"]) < 1e-9
    g.close()


def test_frozen_benchmark_trajectories_are_complete():
    """tests/golden/bench_costs.json (tests/golden/make_bench_cost.py) holds what bench.py's `parity` field and the GPU tests read: the oracle's cost after
    0, 1 and 2 Gauss-Newton steps of 400 PCG iterations at 2048^2 and 4096^2, in float and in double; the float-rounding envelope derived from them is
    what the long-horizon tolerance is made of, so it must be a small positive number, and both precisions must start from the same cost."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    G = json.load(open(os.path.join(root, "tests", "golden", "bench_costs.json")))
    sys.path.insert(0, root)
    import bench
    for size in (2048, 4096):
        f = G[f"image_warping_{size}x{size}_float_gaussNewtonGPU_400"]["costs"]
        d = G[f"image_warping_{size}x{size}_double_gaussNewtonGPU_400"]["costs"]
        assert len(f) == len(d) == 3 and f[0] == d[0] and f[2] < f[1] < f[0] and d[2] < d[1] < d[0]
        gold, env = bench.golden_cost(size, 400)
        assert gold == f and env[0] == 0.0 and all(1e-5 < e < 5e-2 for e in env[1:])
    assert bench.golden_cost(1234, 400) == (None, None)
