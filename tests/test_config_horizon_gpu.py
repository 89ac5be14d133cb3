"""GPU parity tests (-m gpu): BASELINE configs 3 and 4 at their OWN horizons, step by step against frozen oracle cost histories.

Rounds 1-5 compared these configs with the oracle over ONE outer step x 10 PCG iterations (tests/test_steady_state_gpu.py) and checked the full runs only through
properties (tests/test_fullsize_gpu.py).  Here the oracle's whole trajectory is frozen offline (tests/golden/make_config_costs.py -> config_costs.json; config 3 also under
three other legal roundings, make_config3_legal_runs.py -> config3_legal_runs.json) and BOTH HIP paths -- the default loops and the reference-ordered loop
(Opt_SetSolverParameter amd_reference_order = 1) -- are stepped side by side with it (tools/config_horizon.py):

  config 4  arap_mesh_deformation 708 x 707 (500 556 vertices) float, Gauss-Newton 20 x 100 (examples/arap_mesh_deformation/src/main.cpp:75-79)
            The 1e-5 contract holds at EVERY one of the 20 outer steps, on both paths (measured: at most 1.9e-6 / 1.5e-6, growing ~1.1-1.2x per step).
  config 3  shape_from_shading 1024^2 double, Levenberg-Marquardt 60 x 10 (examples/shape_from_shading/src/main.cpp:27-38)
            1e-12 holds for the first 6 outer steps; from step 7 on the distance grows ~3-3.5x per step -- on the default path AND on the reference-ordered loop alike --
            peaks at 6e-3 / 9e-3 around step 20 and ends at 1.7e-4 / 1.2e-3 after 60 steps: the 60-step LM trajectory itself (trust-region radius ~1e4: nearly undamped
            steps on a non-convex shading term) amplifies last-bit differences, it is not a property of a kernel.  The yardstick that says so is the reference's own
            arithmetic: the same oracle under three other legal roundings (fused-multiply-add build; its own per-warp atomic sums under two seeds) leaves the plain run
            the same way.  Asserted: (i) the contract while it is meaningful (steps 1-6); (ii) per step, at most 10 x the measured distance (frozen bars,
            tests/golden/config_horizon_bars.json); (iii) per step, at most 4 x the largest distance of a legal oracle run from the plain one (measured: at most 2.3 x at step 22 / 1.8 x at step 23,
            below 1 x at most steps -- the HIP paths move with the spread of the reference's own arithmetic over all 60 steps);
            (iv) the same number of accepted / rejected steps and the same final cost to 1 %.
"""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
pytestmark = pytest.mark.gpu


def _load(name):
    p = os.path.join(HERE, "golden", name)
    if not os.path.exists(p):
        pytest.skip(f"{name} not generated")
    return json.load(open(p))


@pytest.fixture(scope="module")
def runs():
    import config_horizon as ch
    G = _load("config_costs.json")
    return {c: ch.compare(c, G) for c in (3, 4) if ch.CASES[c][0] in G}


@pytest.mark.parametrize("path", ["default", "reference_order"])
def test_config4_arap_500k_gn_20x100_full_run(runs, path):
    r = runs[4]["paths"][path]
    print(f"config 4 {path}: rel err per step {['%.1e' % e for e in r['rel_err_per_step']]}")
    assert r["same_step_count_as_oracle"] and r["steps"] == 20
    assert r["max_rel_err"] <= 1e-5, r["rel_err_per_step"]      # the float contract at every outer step of the config's own horizon (2000 PCG iterations)


@pytest.mark.parametrize("path", ["default", "reference_order"])
def test_config3_sfs_1024_double_lm_60x10_full_run(runs, path):
    r = runs[3]["paths"][path]
    e = r["rel_err_per_step"]
    print(f"config 3 {path}: first step outside 1e-12: {r['first_step_outside_contract']}, growth per step {r['growth_per_step_until_max']}, max {r['max_rel_err']:.1e}, final {r['final_rel_err']:.1e}")
    assert r["same_step_count_as_oracle"] and r["steps"] == 60      # every accept / reject decision of the 60 LM steps agrees
    assert max(e[:7]) <= 1e-12, e[:7]                               # (i) the contract over the first six outer steps
    assert max(r["radius_rel_err_per_step"][:7]) <= 1e-12
    bars = _load("config_horizon_bars.json").get(f"config3_sfs_1024_double_LM_60x10|{path}")
    if bars:                                                        # (ii) frozen per-step bars: max(contract, 10 x measured)
        for i, (x, b) in enumerate(zip(e, bars["bar_per_step"])):
            assert x <= b, (i, x, b)
    assert e[-1] <= 1e-2                                            # (iv)
    L = _load("config3_legal_runs.json")
    G = _load("config_costs.json")["config3_sfs_1024_double_LM_60x10"]["costs"]
    legal = [v["costs"] for k, v in L.items() if k.startswith("config3_sfs_1024_double_LM_60x10|") and len(v["costs"]) == len(G)]
    if len(legal) >= 2:                                             # (iii) the reference's own arithmetic as the yardstick
        for i in range(7, len(G)):
            yard = max(abs(c[i] - G[i]) / abs(G[i]) for c in legal)
            assert e[i] <= max(1e-12, 4.0 * yard), (i, e[i], yard)      # measured: at most 2.3 x (default path, step 22) / 1.8 x (reference-ordered, step 23) the legal runs' own distance, below it at most steps
