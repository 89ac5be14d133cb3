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
