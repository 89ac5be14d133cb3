"""CPU tests of the input-format readers (opt_amd/io.py).  The round trips always run; the checks against the
reference's own fixture files run only where /root/reference exists (never on the GPU box)."""
import os

import numpy as np
import pytest

from opt_amd import io, workloads as wl

REF = "/root/reference/examples/data"


def test_imagedump_roundtrip(tmp_path):
    a = np.random.default_rng(0).standard_normal((7, 5)).astype(np.float32)
    a[2, 3] = -np.inf; a[0, 0] = np.inf
    io.write_imagedump(tmp_path / "a.imagedump", a)
    np.testing.assert_array_equal(io.read_imagedump(tmp_path / "a.imagedump"), a)
    c = io.read_imagedump(tmp_path / "a.imagedump", clamp_infinity=True)
    assert c[2, 3] == -10000.0 and c[0, 0] == np.finfo(np.float32).max
    m = (np.arange(2 * 7 * 5) % 3).astype(np.uint8).reshape(14, 5)
    io.write_imagedump(tmp_path / "m.imagedump", m)
    np.testing.assert_array_equal(io.read_imagedump(tmp_path / "m.imagedump"), m)
    rgb = np.random.default_rng(1).random((4, 6, 3)).astype(np.float32)
    io.write_imagedump(tmp_path / "c.imagedump", rgb)
    np.testing.assert_array_equal(io.read_imagedump(tmp_path / "c.imagedump"), rgb)


def test_sfs_parameters_roundtrip(tmp_path):
    p = dict(weightFitting=100.0, weightRegularizer=100.0, weightPrior=0.0, weightShading=1.0, weightShadingStart=0.0, weightShadingIncrement=0.0,
             weightBoundary=0.0, fx=574.0, fy=574.5, ux=320.0, uy=240.0, lightingCoefficients=[0.1 * i for i in range(9)])
    io.write_sfs_parameters(tmp_path / "p.SFSSolverParameters", p)
    assert os.path.getsize(tmp_path / "p.SFSSolverParameters") == 160
    q = io.read_sfs_parameters(tmp_path / "p.SFSSolverParameters")
    assert all(abs(q[k] - p[k]) < 1e-6 for k in ("weightFitting", "weightShading", "fx", "fy", "ux", "uy"))
    np.testing.assert_allclose(q["lightingCoefficients"], p["lightingCoefficients"], rtol=1e-6)


def test_constraints_reader(tmp_path):
    (tmp_path / "c.constraints").write_text("2\n30 132 59 44\n229 51 157 91\n")
    assert io.read_constraints(tmp_path / "c.constraints") == [(30, 132, 59, 44), (229, 51, 157, 91)]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference data not present (GPU box)")
def test_reference_fixtures_decode_as_documented():
    assert io.read_constraints(os.path.join(REF, "cat512.constraints")) == wl.CAT512_MARKERS
    p = io.read_sfs_parameters(os.path.join(REF, "shape_from_shading", "default.SFSSolverParameters"))
    fx = wl.SFS_FIXTURE
    assert abs(p["weightFitting"] - fx["w_p"]) < 1e-4 and abs(p["weightRegularizer"] - fx["w_s"]) < 1e-4 and abs(p["weightShading"] - fx["w_g"]) < 1e-4
    assert abs(p["fx"] - fx["fx"]) < 1e-3 and abs(p["ux"] - fx["ux"]) < 1e-3
    np.testing.assert_allclose(p["lightingCoefficients"], fx["L"], atol=1e-4)
    P = io.load_sfs_example(os.path.join(REF, "shape_from_shading", "default"))
    assert P.dims == (640, 480) and P.params[16].shape == (480, 640) and P.params[19].dtype == np.uint8
    assert int((P.params[17] > 0).sum()) == 640 * 480 - 115038          # 115 038 invalid depths


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference data not present (GPU box)")
def test_oracle_runs_on_the_reference_sfs_fixture(oracle_lib):
    P = io.load_sfs_example(os.path.join(REF, "shape_from_shading", "default"))
    o = oracle_lib.OracleSolver("shape_from_shading", "LMGPU", True, P.dims)
    o.set("nIterations", 2); o.set("lIterations", 5)
    o.init(P.params); c0 = o.cost()
    while o.step(P.params):
        pass
    assert np.isfinite(c0) and o.cost() < c0
