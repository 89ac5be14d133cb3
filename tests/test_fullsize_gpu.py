"""GPU tests (-m gpu) at BASELINE.json's full configuration sizes, through size-independent properties (the oracle
would take minutes there): symmetry and positivity of J^T J, its diagonal against diag(J^T J) from evalJTF, cost
decrease under GN / LM, and agreement between the float and double solvers on the same inputs."""
import numpy as np
import pytest

from opt_amd import api, workloads as wl
from helpers import hip_solver

pytestmark = pytest.mark.gpu


def _operator_properties(P, rtol):
    import torch
    g = hip_solver(P)
    dev = api.to_device(P)
    dt = torch.float64 if P.double else torch.float32
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    f, diag = g.eval_jtf(dev)
    act = (diag != 0).to(dt)                     # rows of excluded unknowns have zero diagonal and must stay zero
    p = torch.randn(g.n, device="cuda", generator=gen, dtype=dt) * act
    q = torch.randn(g.n, device="cuda", generator=gen, dtype=dt) * act
    Ap, pAp = g.apply_jtj(dev, p)
    Aq, _ = g.apply_jtj(dev, q)
    pAq, qAp = float(p.double() @ Aq.double()), float(q.double() @ Ap.double())
    assert abs(pAq - qAp) <= rtol * max(abs(pAq), abs(qAp), 1.0)
    assert pAp >= 0 and abs(pAp - float(p.double() @ Ap.double())) <= rtol * max(pAp, 1.0)
    idx = torch.nonzero(act).reshape(-1)
    for i in idx[:: max(1, len(idx) // 5)][:5].tolist():
        e = torch.zeros(g.n, device="cuda", dtype=dt); e[i] = 1
        Ae, _ = g.apply_jtj(dev, e)
        assert abs(float(Ae[i]) - float(diag[i])) <= rtol * abs(float(diag[i]))
    g.close()


def test_config3_sfs_1024_double_lm():
    P = wl.shape_from_shading(1024, 1024, double=True, holes=True)
    _operator_properties(P, 1e-10)
    g = hip_solver(P, "LMGPU", nIterations=6, lIterations=10)
    dev = api.to_device(P)
    g.init(dev); c = [g.cost()]
    while g.step(dev):
        c.append(g.cost())
    assert c[-1] < c[0] and all(b <= a * (1 + 1e-12) for a, b in zip(c, c[1:]))     # LM never accepts an increase
    g.close()


def test_config4_arap_500k():
    P = wl.arap_mesh_deformation(708, 707)
    assert 480_000 < P.dims[0] < 520_000 and P.meta["n_edges"] > 2_900_000
    _operator_properties(P, 2e-4)
    g = hip_solver(P, nIterations=3, lIterations=30)
    dev = api.to_device(P)
    g.init(dev); c = [g.cost()]
    while g.step(dev):
        c.append(g.cost())
    assert len(c) == 4 and c[-1] < 0.9 * c[0]
    g.close()


def test_config1_poisson_256_and_float_vs_double():
    costs = {}
    for double in (False, True):
        P = wl.poisson_image_editing(256, 256, double=double)
        g = hip_solver(P, nIterations=1, lIterations=10)
        dev = api.to_device(P)
        g.init(dev); c0 = g.cost()
        while g.step(dev):
            pass
        costs[double] = (c0, g.cost())
        g.close()
    assert costs[True][1] < costs[True][0]
    assert abs(costs[False][1] - costs[True][1]) <= 1e-5 * costs[True][1]        # float solver within the float bar of the double solver
