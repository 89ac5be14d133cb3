"""GPU parity tests (-m gpu) of the Levenberg-Marquardt inner-loop controls: `residual_reset_period` and `q_tolerance`.

The reference's LM loop (solverGPUGaussNewton.t:1056-1103) ends a PCG iteration either with PCGStep2 or -- every residual_reset_period-th time -- with
the split residual reset (:1077-1083), and leaves the loop when zeta = k (Q_k - Q_{k-1}) / Q_k drops below q_tolerance (:1093-1102).  The HIP loops move
the host's part of this off the GPU's critical path: Q arrives as tagged words one launch late, the next launch *or the reset* is enqueued before Q is
known (and dropped on an early-out: delta is double-buffered), and work whose results die with the loop is not run.  Every combination below is stepped
side by side with the CPU oracle, which follows the reference's order literally:
  * reset periods 1, 2, 3, 5, 7 and the default 10 against lIterations 9 / 10 / 12 (reset on the last iteration, mid-loop restarts, none);
  * q_tolerance at the default, at values that end the loop after a few iterations -- on a reset iteration and next to one -- and at 0 (never);
  * image_warping float + double (single-kernel LM loop of energy_image_warping.hip), shape_from_shading double (energy_sfs.hip), and
    poisson_image_editing (no single-kernel LM loop: the generic Step1/Step2 path of solver.hip).
Costs after every outer step, the trust-region radius and the final unknowns must agree (double 1e-10 / 1e-8 / 1e-9; float 1e-5 on costs).
"""
import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import assert_close, device_unknowns, flat_unknowns, hip_solver, oracle_solver, rel_err

pytestmark = pytest.mark.gpu


def _side_by_side(oracle_lib, P, nsteps, liters, cost_tol, x_tol, radius_tol, hip_only=None, **controls):
    o = oracle_solver(oracle_lib, P, "LMGPU", nIterations=nsteps, lIterations=liters, **controls)
    g = hip_solver(P, "LMGPU", nIterations=nsteps, lIterations=liters, **controls, **(hip_only or {}))
    dev = api.to_device(P)
    Pref = P.clone()
    o.init(Pref.params); g.init(dev)
    scale = max(abs(o.cost()), 1e-300)
    costs = [(o.cost(), g.cost())]
    while True:
        a, b = o.step(Pref.params), g.step(dev)
        assert a == b, (a, b, costs)
        costs.append((o.cost(), g.cost()))
        assert_close("cost" if len(costs) <= 2 else "cost_later", g.cost(), o.cost(), cost_tol, floor=1e-12 * scale, double=P.double, step=len(costs) - 1)
        assert_close("radius", g.trust_region_radius(), o.trust_region_radius(), radius_tol, double=P.double)
        if not a:
            break
    if x_tol is not None:
        assert_close("x", rel_err(device_unknowns(P, dev), flat_unknowns(Pref)), 0.0, x_tol, absolute=True, double=P.double)
    g.close(); o.close()
    return costs


PERIODS = [1, 2, 3, 5, 7, 10]
QTOLS = [None, 0.0, 0.05, 0.5, 5.0]      # None: the default 1e-4;  0.5 / 5: the loop ends within the first iterations (zeta ~ 1/k early on)


@pytest.mark.parametrize("liters", [9, 10, 12])
@pytest.mark.parametrize("period", PERIODS)
def test_image_warping_double_reset_periods(oracle_lib, period, liters):
    P = wl.image_warping(61, 47, double=True, random_state=5, mask_fraction=0.05, perturb=0.4)
    _side_by_side(oracle_lib, P, 4, liters, 1e-10, 1e-9, 1e-8, residual_reset_period=period)


@pytest.mark.parametrize("qtol", QTOLS)
@pytest.mark.parametrize("period", [1, 2, 3, 10])
def test_image_warping_double_q_early_out(oracle_lib, period, qtol):
    P = wl.image_warping(53, 38, double=True, random_state=7, mask_fraction=0.05, perturb=0.4)
    kw = dict(residual_reset_period=period)
    if qtol is not None:
        kw["q_tolerance"] = qtol
    _side_by_side(oracle_lib, P, 4, 12, 1e-10, 1e-9, 1e-8, **kw)


@pytest.mark.parametrize("period,qtol", [(2, None), (3, 0.5), (10, 0.05), (5, 0.0)])
def test_image_warping_float_controls(oracle_lib, period, qtol):
    P = wl.image_warping(64, 48, double=False, random_state=9, mask_fraction=0.05, perturb=0.4)
    kw = dict(residual_reset_period=period)
    if qtol is not None:
        kw["q_tolerance"] = qtol
    _side_by_side(oracle_lib, P, 3, 10, 1e-5, None, 1e-3, **kw)


@pytest.mark.parametrize("double", [True, False])
@pytest.mark.parametrize("period,qtol,liters", [(1, None, 6), (2, None, 10), (3, None, 12), (3, 0.5, 10), (10, None, 25), (10, 0.05, 12), (4, 0.0, 9), (7, None, 23), (5, 5.0, 10),
                                                (10, 0.01, 40), (6, 0.002, 60), (10, None, 1), (10, None, 2), (10, None, 3), (2, None, 3), (1, None, 1), (1, None, 2), (3, 5.0, 4)])
def test_image_warping_launch_per_iteration_loop_controls(oracle_lib, period, qtol, liters, double):
    """The same controls on the launch-per-iteration LM loop ("amd_onchip" = 0: what images past the on-chip range run on -- the reference's LM-only large-image case,
    examples/image_warping/src/main.cpp:121-129).  Round 6: on a unit lattice that loop keeps no residual vector (ring of three p buffers; true r only behind PCGInit1 and
    behind a reset) and takes Q from the CG recurrence Q_k = Q_{k-1} + alpha_k (p_k . r_k - 1/2 alpha_k p_k . A p_k) instead of summing 1/2 delta . (r + b); a reset
    re-anchors it to the direct sum.  Early-outs on, next to and between resets, restarts, long solves -- step for step beside the oracle, which sums Q directly."""
    P = wl.image_warping(61, 47, double=double, random_state=5, mask_fraction=0.05, perturb=0.4)
    kw = dict(residual_reset_period=period)
    if qtol is not None:
        kw["q_tolerance"] = qtol
    if double:
        _side_by_side(oracle_lib, P, 4, liters, 1e-10, 1e-9, 1e-8, hip_only=dict(amd_onchip=0), **kw)
    else:
        _side_by_side(oracle_lib, P, 3, liters, 1e-5, None, 1e-3, hip_only=dict(amd_onchip=0), **kw)
    g = hip_solver(P, "LMGPU", timing=True, nIterations=1, lIterations=liters, amd_onchip=0, **kw)
    dev = api.to_device(P)
    g.init(dev); g.step(dev)
    kt = g.kernel_timings()
    g.close()
    assert "PCGIteration" in kt and "PCGSolveOnChip" not in kt, kt.keys()


@pytest.mark.parametrize("period,qtol,liters", [(1, None, 6), (2, None, 10), (3, 0.5, 10), (10, None, 10), (10, 0.05, 12), (4, 0.0, 9), (5, 5.0, 10)])
def test_sfs_double_controls(oracle_lib, period, qtol, liters):
    P = wl.shape_from_shading(72, 56, double=True, seed=2)
    kw = dict(residual_reset_period=period)
    if qtol is not None:
        kw["q_tolerance"] = qtol
    _side_by_side(oracle_lib, P, 4, liters, 1e-10, 1e-9, 1e-8, **kw)


@pytest.mark.parametrize("period,qtol,liters", [(2, None, 10), (10, None, 10), (3, 0.5, 9), (10, 0.0, 12)])
def test_poisson_generic_lm_loop_controls(oracle_lib, period, qtol, liters):
    P = wl.poisson_image_editing(40, 36, seed=4)
    kw = dict(residual_reset_period=period)
    if qtol is not None:
        kw["q_tolerance"] = qtol
    _side_by_side(oracle_lib, P, 3, liters, 1e-5, None, 1e-3, **kw)


@pytest.mark.parametrize("double", [True, False])
@pytest.mark.parametrize("period,qtol,liters", [(1, None, 6), (2, None, 10), (3, 0.5, 10), (10, None, 10), (10, 0.05, 12), (4, 0.0, 9), (7, None, 23), (5, 5.0, 10)])
def test_arap_two_kernel_lm_iteration_controls(oracle_lib, period, qtol, liters, double):
    """arap_mesh_deformation, Levenberg-Marquardt on the record gather (round 6: arap_flatStepRec<.., LM> + arap_applySym with CtC -- two kernels per PCG iteration where
    the generic loop runs three; reference shape: examples/arap_mesh_deformation/src/main.cpp:81-99): CtC in the gather, b / Q / deltaOut in the flat pass, the restart
    launch after a split residual reset, early-outs on and next to a reset iteration.  The two-kernel loop must actually be the one that ran (kernel names)."""
    P = wl.arap_mesh_deformation(36, 29, double=double, perturb=0.01)
    kw = dict(residual_reset_period=period)
    if qtol is not None:
        kw["q_tolerance"] = qtol
    if double:
        _side_by_side(oracle_lib, P, 4, liters, 1e-10, 1e-9, 1e-8, **kw)
    else:
        _side_by_side(oracle_lib, P, 3, liters, 1e-5, None, 1e-3, **kw)
    g = hip_solver(P, "LMGPU", timing=True, nIterations=1, lIterations=max(liters, 3), residual_reset_period=period)
    dev = api.to_device(P)
    g.init(dev); g.step(dev)
    kt = g.kernel_timings()
    g.close()
    assert "PCGStep2+PCGStep3" in kt and "PCGStep3" not in kt, kt.keys()


def test_verbose_run_takes_the_same_path_as_the_silent_one(oracle_lib, capfd):
    """Round 6: verbosity > 0 no longer sends an LM solve to the launch-per-iteration loop -- the on-chip kernel reports the iteration and zeta of its q
    early-out in a pinned word and the solver prints the reference's "breaking at iteration" message from it after the step has drained.  Same path => the SAME bits as the
    silent run, and the message appears when an early-out happens (q_tolerance 0.5 ends every linear solve after a few iterations)."""
    P = wl.image_warping(48, 40, double=True, random_state=3, perturb=0.3)
    silent = _side_by_side(oracle_lib, P, 3, 10, 1e-10, 1e-9, 1e-8, q_tolerance=0.5)
    capfd.readouterr()
    g = hip_solver(P, "LMGPU", verbosity=1, nIterations=3, lIterations=10, q_tolerance=0.5)
    dev = api.to_device(P)
    g.init(dev)
    costs = [g.cost()]
    while g.step(dev):
        costs.append(g.cost())
    assert g.on_chip_status() == 1
    g.close()
    import ctypes
    ctypes.CDLL(None).fflush(None)      # the library prints through C stdio
    out = capfd.readouterr().out
    assert costs == [c[1] for c in silent][:len(costs)], (costs, silent)
    assert "breaking at iteration" in out, out[-2000:]
