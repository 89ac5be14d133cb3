#!/usr/bin/env python
"""Time BASELINE.json's configs 1-4 on one MI355X (the metric itself, config 5 / image_warping 4096^2, is bench.py).

    python tools/bench_configs.py > gpurun_out/configs.json

One JSON line per config: solver wall time (Opt_ProblemInit + all Opt_ProblemStep calls, inputs resident in HBM), PCG
iterations/s, final cost, and per-kernel hipEvent averages from a second, timed solve.  The iteration counts are
the reference harness defaults.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402
from opt_amd import api, workloads as wl       # noqa: E402

CONFIGS = [
    ("config1 poisson_image_editing 256x256 float GN 1x10", lambda: wl.poisson_image_editing(256, 256), "gaussNewtonGPU", 1, 10),
    ("poisson_image_editing 2048x2048 float GN 1x100", lambda: wl.poisson_image_editing(2048, 2048), "gaussNewtonGPU", 1, 100),
    ("config2 image_warping 2048x2048 float GN 8x400", lambda: wl.image_warping(2048, 2048), "gaussNewtonGPU", 8, 400),
    ("config3 shape_from_shading 1024x1024 double LM 60x10", lambda: wl.shape_from_shading(1024, 1024, double=True), "LMGPU", 60, 10),
    ("config4 arap_mesh_deformation 708x707 grid (500k vertices) float GN 20x100", lambda: wl.arap_mesh_deformation(708, 707), "gaussNewtonGPU", 20, 100),
    ("image_warping 2048x2048 float LM 8x400", lambda: wl.image_warping(2048, 2048), "LMGPU", 8, 400),
    # the reference's ARAP performance run is GN and LM with 1000 linear iterations (arap_mesh_deformation/src/main.cpp:81-99); LM takes the two-kernel iteration since round 6
    ("arap_mesh_deformation 708x707 grid (500k vertices) float LM 3x1000", lambda: wl.arap_mesh_deformation(708, 707), "LMGPU", 3, 1000),
    # the functor-engine energies at their examples' iteration counts (not BASELINE configs)
    ("extra optical_flow 1024x1024 float GN 3x50", lambda: wl.optical_flow(1024, 1024), "gaussNewtonGPU", 3, 50),
    ("extra intrinsic_image_decomposition 1024x1024 float GN 7x10", lambda: wl.intrinsic_image_decomposition(1024, 1024), "gaussNewtonGPU", 7, 10),
    ("extra volumetric_mesh_deformation 96^3 float GN 20x60", lambda: wl.volumetric_mesh_deformation(96, 96, 96), "gaussNewtonGPU", 20, 60),
    ("extra cotangent_mesh_smoothing 512x512 torus float GN 5x25", lambda: wl.cotangent_mesh_smoothing(512, 512), "gaussNewtonGPU", 5, 25),
    ("extra embedded_mesh_deformation 512x512 float GN 5x125", lambda: wl.embedded_mesh_deformation(512, 512), "gaussNewtonGPU", 5, 125),
    ("extra robust_nonrigid_alignment 512x512 float GN 5x50", lambda: wl.robust_nonrigid_alignment(512, 512), "gaussNewtonGPU", 5, 50),
    # the block-local patch solver: 4 outer steps x 16 sweeps of 16 in-LDS PCG iterations, 32x32 patches
    ("extra poisson_image_editing 2048x2048 float patch solver 4x16", lambda: wl.poisson_image_editing(2048, 2048), "patchGaussNewtonGPU", 4, 16),
]


def run(P, kind, nit, lit, timing):
    dev = api.to_device(P)
    s = api.Solver(api.energy_file(P.energy), kind, P.dims, double=P.double, timing=timing)
    s.set_parameter("nIterations", nit); s.set_parameter("lIterations", lit)
    if timing or kind == "LMGPU":
        s.enable_trace(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.init(dev)
    c0 = s.cost(); steps = 0
    while s.step(dev):
        steps += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = s.kernel_timings() if timing else None
    global LAST_PLAN
    LAST_PLAN = {"describe": s.describe(), "on_chip_status": s.on_chip_status()}      # which linear-solve path the plan took (OptAmd_PlanDescribe / OptAmd_PlanOnChipStatus)
    out = (dt, c0, s.cost(), steps, kt)
    s.close()
    return out


LAST_PLAN = {}


def cpu_port(P, kind, lit):
    """The CPU oracle (a port of the reference algorithm, not the reference) on a bounded sample of the same problem: one outer
    iteration with at most `lit` PCG iterations, all host cores for the row-banded image energies."""
    from oracle.binding import OracleSolver
    threads = max(1, min(os.cpu_count() or 1, 128))
    o = OracleSolver(P.energy, kind, P.double, P.dims)
    o.set_threads(threads)
    o.set("nIterations", 1); o.set("lIterations", lit)
    Q = P.clone()
    o.init(Q.params)
    t0 = time.perf_counter()
    o.step(Q.params)
    dt = time.perf_counter() - t0
    o.close()
    return {"pcg_iters_per_s": lit / dt, "wall_s": dt, "threads": threads, "sample": f"1 outer iteration x {lit} PCG iterations (incl. that step's J^T F, update and cost)"}


def main():
    only = os.environ.get("OPT_AMD_CONFIG")          # substring filter, e.g. "config3" (used by tools/profile_config.sh)
    with_cpu = os.environ.get("OPT_AMD_CPU_PORT") == "1"
    for name, make, kind, nit, lit in CONFIGS:
        if only and only not in name:
            continue
        P = make()
        run(make(), kind, 1, min(lit, 5), False)                     # warm-up (module load, allocator)
        dt, c0, c1, steps, _ = run(P, kind, nit, lit, False)
        kt = None
        if os.environ.get("OPT_AMD_NO_TIMING_RUN") != "1":               # tools/timeline_gaps.py wants the plain solve last in the trace
            _, _, _, _, kt = run(make(), kind, min(nit, 3), lit, True)
        pcg = sum(v[0] for k, v in kt.items() if k in ("PCGStep2", "PCGStep2_2ndHalf", "PCGIteration")) if kt else 0
        row = {"config": name, "solver": kind, "double": P.double, "wall_s": dt, "outer_steps": steps, "cost_initial": c0, "cost_final": c1,
               "pcg_iters_per_s_nominal": steps * lit / dt, "kernel_avg_us": {k: round(v[1] / v[0] * 1e3, 2) for k, v in (kt or {}).items()},
               "pcg_iterations_in_timed_solve": pcg, "linear_solve_launches_in_timed_solve": (kt or {}).get("PCGSolveOnChip", (0, 0))[0], "plan": LAST_PLAN}
        if with_cpu and not kind.startswith("patch"):
            row["cpu_port"] = cpu_port(make(), kind, min(lit, 10))
            row["gpu_over_cpu_port"] = row["pcg_iters_per_s_nominal"] / row["cpu_port"]["pcg_iters_per_s"]
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
