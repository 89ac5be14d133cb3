"""GPU tests (-m gpu): co-residency of the persistent ("on-chip") kernels is a requirement the runtime CHECKS.

The on-chip linear solves (iw_onchipPcg, sfs_onchipPcg, march_onchipPcg) wait for each other's words, so their whole grid must be resident.  Three mechanisms
(opt_amd/csrc/solver.hip "co-residency", onchip_sync.h ocTimeouts):
  (i)   a process-wide lease per device serialises the on-chip launches of plans stepped from different host threads;
  (ii)  the waits of a launch's FIRST phase are bounded by 10 ms: passing them proves the grid resident, nothing has been written before; a foreign tenant that
        holds CUs makes the launch give up there, the step is redone by the streaming kernels;
  (iii) after such a fall-back the plan returns to the chip after 8 (16, 32 ...) clean steps.
Checked here: two plans (image_warping 512^2 GN and shape_from_shading 640x480 double LM) stepped concurrently from two threads give, each, exactly the costs they give
alone, stay on chip, and no step stalls; a 1024^2 plan (256 workgroups, one per CU) stepped while a foreign kernel holds half the CUs for 0.3 s falls back ONCE (status 2, oracle-correct cost, stall << 50 ms
beyond its own work) and is back on chip (status 1) after its back-off.
"""
import ctypes
import threading
import time

import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import hip_solver, oracle_solver

pytestmark = pytest.mark.gpu


def _stepper(P, kind, nsteps, liters, barrier=None, pause=0.0, **params):
    import torch
    g = hip_solver(P, kind, nIterations=nsteps, lIterations=liters, **params)
    dev = api.to_device(P)
    g.init(dev)
    torch.cuda.synchronize()
    if barrier is not None:
        barrier.wait()
    costs, status, times = [g.cost()], [], []
    while True:
        t0 = time.perf_counter()
        more = g.step(dev)
        times.append(time.perf_counter() - t0)
        if not more:
            break
        costs.append(g.cost()); status.append(g.on_chip_status())
        if pause:
            time.sleep(pause)
    g.close()
    return costs, status, times


def test_two_plans_stepped_from_two_threads():
    PA = wl.image_warping(512, 512)
    PB = wl.shape_from_shading(640, 480, double=True, seed=1)
    alone = {"A": _stepper(PA, "gaussNewtonGPU", 40, 50), "B": _stepper(PB, "LMGPU", 40, 10)}
    assert all(s == 1 for s in alone["A"][1]) and all(s == 1 for s in alone["B"][1])
    res = {}
    bar = threading.Barrier(2)
    ta = threading.Thread(target=lambda: res.update(A=_stepper(PA, "gaussNewtonGPU", 40, 50, bar)))
    tb = threading.Thread(target=lambda: res.update(B=_stepper(PB, "LMGPU", 40, 10, bar)))
    ta.start(); tb.start(); ta.join(); tb.join()
    for k in "AB":
        costs, status, times = res[k]
        # the lease makes every step take the path it takes alone: the SAME bits (a step that could not get the chip within 50 ms would run on the streaming kernels and
        # show as status 0 and a cost ~1e-7 away; none does)
        assert status == alone[k][1], (k, status)
        assert costs == alone[k][0], (k, costs, alone[k][0])
        t = sorted(times[1:])
        assert t[int(0.9 * (len(t) - 1))] < 0.05 and t[-1] < 0.5, (k, t[-3:])      # no stall: the other plan's linear solve is at most a few ms (one outlier of the host's scheduler is not the GPU's)


def test_foreign_tenant_makes_the_launch_fall_back_once_and_the_plan_returns(oracle_lib):
    import torch
    P = wl.image_warping(1024, 1024)      # ROWS = 8 variant: 256 tiles of 256 x 16 pixels, 246 VGPRs -- one workgroup per CU, all 256 CUs needed
    o = oracle_solver(oracle_lib, P, "gaussNewtonGPU", nIterations=3, lIterations=10)
    o.set_threads(8)
    Pref = P.clone()
    o.init(Pref.params)
    oc = [o.cost()]
    while o.step(Pref.params):
        oc.append(o.cost())
    o.close()
    # what the fall-back must reproduce BIT FOR BIT: step 1 on chip, steps 2 and 3 on the streaming kernels (amd_onchip = 0 from step 2 on) -- undisturbed
    ref = hip_solver(P, "gaussNewtonGPU", nIterations=3, lIterations=10)
    dref = api.to_device(P)
    ref.init(dref); ref.step(dref)
    ref.set_parameter("amd_onchip", 0)
    mixed = []
    while ref.step(dref):
        mixed.append(ref.cost())
    ref.close()

    g = hip_solver(P, "gaussNewtonGPU", nIterations=14, lIterations=10)
    dev = api.to_device(P)
    g.init(dev)
    g.step(dev)                                      # step 1 alone: on chip
    assert g.on_chip_status() == 1
    side = torch.cuda.Stream()
    # 128 one-wave workgroups with 150 KB of LDS each: half the CUs are closed to the plan's workgroups.  (A tenant that fills a whole shader engine -- 240 of these --
    # blocks EVERY kernel's workgroups assigned to that engine until it leaves, persistent or not: measured 280 ms for this step; nothing a solver can do about that.)
    assert api.lib().OptAmd_DebugOccupy(128, ctypes.c_double(300.0), ctypes.c_void_p(side.cuda_stream)) == 1
    time.sleep(0.02)                                 # the tenant is running
    t0 = time.perf_counter()
    g.step(dev)                                      # step 2 while 128 CUs are held: first-phase wait gives up after 10 ms, redone on the streaming kernels
    dt = time.perf_counter() - t0
    assert g.on_chip_status() == 2, g.describe()
    assert dt < 0.1, dt                              # 10 ms bound + the streaming redo (measured 10.6 ms); the old 2 s time-out -- or waiting for the tenant's 300 ms -- would show here
    costs = [g.cost()]
    g.step(dev); costs.append(g.cost())
    assert costs == mixed[:2], (costs, mixed)      # the redone step and the next one: the same bits as an undisturbed run that takes the streaming kernels from step 2 on
    # (and the oracle's to the accuracy this input allows after 10 PCG iterations: its cost has fallen 2500-fold from 8.9e6, every HIP loop -- the reference-ordered one
    # included -- ends 3e-3 from the plain oracle build and within 4e-4 of each other here; the 1e-5 statements are tests/test_steady_state_gpu.py's)
    np.testing.assert_allclose(costs, oc[2:4], rtol=2e-2)
    side.synchronize()
    seen = []
    while g.step(dev):
        seen.append(g.on_chip_status())
    assert 1 in seen and seen[-1] == 1, seen         # back on chip after the back-off (8 clean steps)
    assert g.describe().get("onchip_fallbacks") == "1", g.describe()
    g.close()
