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


def test_mesh_and_marker_readers(tmp_path):
    """OFF / PLY (ascii and binary_little_endian) / MRK readers of the mesh examples, on files written here."""
    import struct
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float32)
    F = [[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]]
    (tmp_path / "t.off").write_text("OFF\n4 4 0\n" + "\n".join(" ".join(str(x) for x in v) for v in V) + "\n" + "\n".join("3 " + " ".join(map(str, f)) for f in F) + "\n")
    hdr = "ply\nformat {}\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nelement face 4\nproperty list uchar int vertex_indices\nend_header\n"
    (tmp_path / "a.ply").write_text(hdr.format("ascii 1.0") + "\n".join(" ".join(str(x) for x in v) for v in V) + "\n" + "\n".join("3 " + " ".join(map(str, f)) for f in F) + "\n")
    (tmp_path / "b.ply").write_bytes(hdr.format("binary_little_endian 1.0").encode() + V.tobytes() + b"".join(struct.pack("<B3i", 3, *f) for f in F))
    (tmp_path / "t.mrk").write_text("2\n0.5 0.5 0.5 0.02 3\n-1 0 0 0.02 0\n")
    for name, reader in (("t.off", io.read_off), ("a.ply", io.read_ply), ("b.ply", io.read_ply)):
        V2, F2 = reader(str(tmp_path / name))
        np.testing.assert_array_equal(V2, V); assert F2 == F
    idx, pos = io.read_mrk(str(tmp_path / "t.mrk"))
    assert idx.tolist() == [3, 0] and pos.tolist() == [[0.5, 0.5, 0.5], [-1, 0, 0]]
    heads, tails = io.mesh_half_edges(4, F)
    assert heads.tolist() == [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3] and tails.tolist() == [1, 2, 3, 0, 2, 3, 0, 1, 3, 0, 1, 2]
    P = io.arap_problem_from_mesh(V, F, idx, pos, double=True, alpha=0.5)
    assert P.dims == (4,) and int(P.params[6]) == 12
    np.testing.assert_allclose(P.params[5][3], [0.25, 0.25, 0.75]); assert np.isneginf(P.params[5][1]).all()


def test_raptor_fixture_is_a_closed_symmetric_mesh():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "raptor2k_mesh.npz"))
    heads, tails = io.mesh_half_edges(len(z["vertices"]), z["faces"].tolist())
    assert len(z["vertices"]) == 2000 and len(z["faces"]) == 4036 and len(heads) == 12108
    fwd = set(zip(heads.tolist(), tails.tolist()))
    assert all((b, a) in fwd for a, b in fwd)                       # every half-edge has its opposite
    deg = np.bincount(heads, minlength=2000)
    assert deg.min() >= 3 and deg.max() == 12 and (np.diff(heads) >= 0).all()   # grouped by head vertex (OptGraph.h:64-76)
    assert z["marker_index"].max() < 2000


def test_png_reader_all_filters(tmp_path):
    """read_png against PNGs written here with each scanline filter type (zlib stream built by hand)."""
    import struct
    import zlib
    rng = np.random.default_rng(0)
    for ch, ctype in ((1, 0), (3, 2), (4, 6)):
        img = rng.integers(0, 256, size=(7, 5, ch), dtype=np.uint8)
        H, W = img.shape[:2]
        rows = img.reshape(H, W * ch).astype(np.int32)
        body = b""
        for y in range(H):
            f = y % 5
            prev = rows[y - 1] if y else np.zeros(W * ch, dtype=np.int32)
            line = np.zeros(W * ch, dtype=np.int32)
            for i in range(W * ch):
                a = rows[y][i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                paeth = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                pred = [0, a, b, (a + b) >> 1, paeth][f]
                line[i] = (rows[y][i] - pred) & 255
            body += bytes([f]) + line.astype(np.uint8).tobytes()

        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(body)) + chunk(b"IEND", b"")
        p = tmp_path / f"f{ch}.png"
        p.write_bytes(png)
        np.testing.assert_array_equal(io.read_png(str(p)), img)


def test_image_fixtures_have_the_expected_content():
    here = os.path.dirname(os.path.abspath(__file__))
    m = np.load(os.path.join(here, "fixtures", "cat_mask_128.npz"))["mask_red"]
    assert m.shape == (128, 128) and set(np.unique(m)) == {0, 255} and 0.3 < (m == 0).mean() < 0.45
    z = np.load(os.path.join(here, "fixtures", "poisson_real_112x80.npz"))
    assert z["base"].shape == (80, 112, 3) and z["inserted"].shape == (80, 112, 3) and set(np.unique(z["mask"])) == {0, 255}
    P = io.image_warping_problem_from_mask(m, downsample=1, double=True)
    assert P.dims == (128, 128) and (np.asarray(P.params[3])[..., 0] >= 0).sum() >= 4 * 127
