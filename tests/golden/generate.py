#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the CPU oracle (double precision).

The reference ships no golden vectors and cannot be executed here (Terra/Lua -> PTX), so these vectors are
NOT reference outputs: they freeze the oracle's answers on small seeded instances so that (a) the oracle itself
is regression-tested on CPU and (b) the HIP path is checked against fixed numbers that do not depend on the
oracle being rebuilt identically.  Each .npz holds the inputs (problem parameters by binding index), the
stage outputs (cost, J^T F, diag(J^T J), J^T J p for a seeded p) and the trajectory of a 3 GN x 10 PCG solve
(cost per GN iteration; alphaNum/alphaDen/betaNum per PCG iteration; final unknowns).

    python tests/golden/generate.py          # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from opt_amd import workloads as wl          # noqa: E402
from oracle.binding import OracleSolver      # noqa: E402

CASES = {
    "image_warping_24x20": (lambda: wl.image_warping(24, 20, double=True, random_state=11, mask_fraction=0.07, perturb=0.35), "gaussNewtonGPU"),
    "poisson_16x12": (lambda: wl.poisson_image_editing(16, 12, double=True, seed=5), "gaussNewtonGPU"),
    "arap_7x6": (lambda: wl.arap_mesh_deformation(7, 6, double=True, seed=8, perturb=0.01), "gaussNewtonGPU"),
    "sfs_20x16_lm": (lambda: wl.shape_from_shading(20, 16, double=True, seed=9, holes=True, noise=2e-3), "LMGPU"),
    "curve_fitting_64": (lambda: wl.curve_fitting(64, double=True), "gaussNewtonGPU"),
    "optical_flow_18x14": (lambda: wl.optical_flow(18, 14, double=True, seed=12, init_flow=0.9), "gaussNewtonGPU"),
    "intrinsic_14x12_lm": (lambda: wl.intrinsic_image_decomposition(14, 12, double=True, seed=13), "LMGPU"),
    "volumetric_5x4x4": (lambda: wl.volumetric_mesh_deformation(5, 4, 4, double=True, seed=14, perturb=0.04), "gaussNewtonGPU"),
    "cotangent_torus_8x6": (lambda: wl.cotangent_mesh_smoothing(8, 6, double=True, seed=15), "gaussNewtonGPU"),
    "embedded_7x5_lm": (lambda: wl.embedded_mesh_deformation(7, 5, double=True, seed=16, perturb=0.03), "LMGPU"),
    "robust_7x6": (lambda: wl.robust_nonrigid_alignment(7, 6, double=True, seed=17, perturb=0.03), "gaussNewtonGPU"),
}


def build(name):
    make, kind = CASES[name]
    P = make()
    out = {"energy": P.energy, "kind": kind, "dims": np.array(P.dims), "unknown_slots": np.array(P.unknown_slots)}
    for i, p in enumerate(P.params):
        out[f"param_{i}"] = np.array(p)
    o = OracleSolver(P.energy, kind, True, P.dims)
    out["cost"] = np.array(o.eval_cost(P.params))
    f, d = o.eval_jtf(P.params)
    out["jtf"], out["diag"] = f, d
    rng = np.random.default_rng(123)
    p = rng.standard_normal(o.n)
    out["p"] = p
    out["jtjp_unmasked"] = o.apply_jtj(P.params, p)     # p also non-zero on excluded rows: pure operator check
    Q = P.clone()
    o.set("nIterations", 3); o.set("lIterations", 10)
    o.solve(Q.params)
    out["cost_history"] = o.cost_history()
    out["trace"] = o.trace()
    out["final_unknowns"] = np.concatenate([np.asarray(Q.params[s]).reshape(-1) for s in Q.unknown_slots])
    return out


def build_patch():
    """Block-local patch solver (oracle/patch.hpp) on a ragged-mask poisson problem: X after every outer step for both patch sizes.
    Lives in patch/ because its schema differs from the per-energy fixtures above."""
    from oracle.binding import poisson_patch_solve
    W, H = 29, 23
    P = wl.poisson_image_editing(W, H, double=True, seed=21)
    X, T, M = [np.array(a) for a in P.params]
    ys, xs = np.mgrid[0:H, 0:W]
    M[:] = 255.0
    M[(xs - 13) ** 2 + (ys - 11) ** 2 < 81] = 0.0
    M[7:9, :] = 0.0
    M[10, 12] = 255.0
    out = {"X": X, "T": T, "M": M, "nIterations": np.array(3), "lIterations": np.array(5), "patchIterations": np.array(16)}
    for ps in (16, 32):
        Xo, costs = poisson_patch_solve(X, T, M, 3, 5, 16, ps)
        out[f"X_final_{ps}"] = Xo
        out[f"costs_{ps}"] = costs
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or (list(CASES) + ["patch"])          # optional: only the named cases
    for name in names:
        if name == "patch":
            os.makedirs(os.path.join(HERE, "patch"), exist_ok=True)
            np.savez_compressed(os.path.join(HERE, "patch", "poisson_patch_29x23.npz"), **build_patch())
        else:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **build(name))
        print("wrote", name)
