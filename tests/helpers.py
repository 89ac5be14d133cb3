"""Shared helpers of the parity tests: run the same problem through the CPU oracle and the HIP library."""
import numpy as np

from opt_amd import api


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.linalg.norm(b), 1e-300)
    return float(np.linalg.norm(a - b) / den)


def flat_unknowns(problem):
    return np.concatenate([np.asarray(problem.params[i]).reshape(-1) for i in problem.unknown_slots])


def oracle_solver(oracle_lib, problem, kind="gaussNewtonGPU", **params):
    s = oracle_lib.OracleSolver(problem.energy, kind, problem.double, problem.dims)
    for k, v in params.items():
        s.set(k, v)
    return s


def hip_solver(problem, kind="gaussNewtonGPU", timing=False, verbosity=0, **params):
    s = api.Solver(api.energy_file(problem.energy), kind, problem.dims, double=problem.double, verbosity=verbosity, timing=timing)
    for k, v in params.items():
        s.set_parameter(k, v)
    return s


def device_unknowns(problem, dev_params):
    import torch
    return torch.cat([dev_params[i].reshape(-1) for i in problem.unknown_slots]).cpu().numpy()


def active_mask(P):
    """Boolean mask over the flat unknown vector: rows of non-excluded unknowns."""
    if P.energy == "image_warping":
        m = np.asarray(P.params[4]).reshape(-1) == 0
        return np.concatenate([np.repeat(m, 2), m])
    if P.energy == "poisson_image_editing":
        return np.repeat(np.asarray(P.params[2]).reshape(-1) == 0, 4)
    if P.energy == "shape_from_shading":
        return np.asarray(P.params[17]).reshape(-1) > 0
    return np.ones(flat_unknowns(P).size, dtype=bool)


# ---- per-case parity bars ----------------------------------------------------------------------------------------
# north_star: 1e-5 (float) / 1e-12 (double).  A double trajectory of k PCG iterations does not hold 1e-12 everywhere -- the iteration amplifies last-bit
# differences of the sums -- so every test FUNCTION carries its own bar per quantity: max(contract, 10 x the largest error measured over all its parametrisations
# on the GPU), frozen in tests/golden/parity_bars.json together with the measured value and the reason (tools/make_parity_bars.py builds the table from a logged
# run: OPT_PARITY_LOG=<file> python -m pytest tests -m gpu).  A function without an entry keeps the default the call site passes (the pre-round-6 blanket bar).
import json as _json
import os as _os

_BARS = None


def _bars():
    global _BARS
    if _BARS is None:
        p = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "parity_bars.json")
        _BARS = _json.load(open(p)) if _os.path.exists(p) else {}
    return _BARS


def _current_test():
    """('tests/test_x.py::test_name', '[params]') of the running test (pytest sets PYTEST_CURRENT_TEST)."""
    node = _os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    fn, _, par = node.partition("[")
    return fn, ("[" + par) if par else ""


def assert_close(kind, got, ref, default_tol, floor=0.0, absolute=False, double=None, step=None):
    """|got - ref| <= bar * max(|ref|, floor)   (absolute: got <= bar).  kind: 'cost0' (before any step), 'cost', 'radius', 'x'.  The bar is the test function's entry in
    tests/golden/parity_bars.json (key '<file>::<function>|<double|float>|<kind>') if there is one, else default_tol.  With OPT_PARITY_LOG set, the measured relative
    error is appended to that file as one JSON line per check (the input of tools/make_parity_bars.py)."""
    fn, par = _current_test()
    prec = "double" if double else "float" if double is not None else "any"
    err = abs(got) if absolute else abs(got - ref) / max(abs(ref), floor, 1e-300)
    log = _os.environ.get("OPT_PARITY_LOG")
    if log:
        with open(log, "a") as f:
            f.write(_json.dumps({"test": fn, "params": par, "prec": prec, "kind": kind, "step": step, "err": err, "default": default_tol}) + "\n")
    e = _bars().get(f"{fn}{par}|{prec}|{kind}") or _bars().get(f"{fn}|{prec}|{kind}")      # (a per-parametrisation entry wins: tests parametrised over ENERGIES)
    tol = e["bar"] if e else default_tol
    assert err <= tol, (kind, step, got, ref, err, tol, "bar from parity_bars.json" if e else "default bar")
