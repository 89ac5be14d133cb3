"""On-disk formats of the reference's example inputs -- host-side readers / writers.

  .imagedump            int32 width, height, channelCount, datatype (0 = float32, 1 = uint8) + row-major payload
                        (reference API/src/im.t:7-53; examples/shape_from_shading/src/SimpleBuffer.cpp:12-52)
  .SFSSolverParameters  the 160-byte TerraSolverParameters struct (examples/shape_from_shading/src/TerraSolverParameters.h:7-44)
  .constraints          text: marker count, then `srcX srcY dstX dstY` per marker (examples/image_warping/src/main.cpp:4-27)
"""
import struct

import numpy as np

from . import workloads as wl


def read_imagedump(path, clamp_infinity=False):
    """-> (H, W) or (H, W, C) array.  clamp_infinity mirrors SimpleBuffer: +inf -> FLT_MAX, -inf -> -10000."""
    with open(path, "rb") as f:
        w, h, c, dt = struct.unpack("<4i", f.read(16))
        if dt == 0:
            a = np.frombuffer(f.read(4 * w * h * c), dtype="<f4").copy()
        elif dt == 1:
            a = np.frombuffer(f.read(w * h * c), dtype=np.uint8).copy()
        else:
            raise ValueError(f"{path}: unknown imagedump datatype {dt}")
    if a.size != w * h * c:
        raise ValueError(f"{path}: truncated payload")
    if dt == 0 and clamp_infinity:
        a[np.isposinf(a)] = np.finfo(np.float32).max
        a[np.isneginf(a)] = -10000.0
    return a.reshape(h, w) if c == 1 else a.reshape(h, w, c)


def write_imagedump(path, a):
    a = np.asarray(a)
    if a.dtype == np.uint8:
        dt = 1
    else:
        a, dt = a.astype("<f4"), 0
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", w, h, c, dt))
        f.write(np.ascontiguousarray(a).tobytes())


_SFS_FIELDS = ["weightFitting", "weightRegularizer", "weightPrior", "weightShading", "weightShadingStart", "weightShadingIncrement",
               "weightBoundary", "fx", "fy", "ux", "uy"]


def read_sfs_parameters(path):
    """-> dict with the named scalars, deltaTransform (4x4) and lightingCoefficients (9)."""
    blob = open(path, "rb").read()
    if len(blob) in (156, 160):   # 11 floats, float4x4 (16 floats), 9 floats, 3 uints; the shipped blob carries 4 trailing pad bytes
        off_m, off_l = 44, 108
    else:
        raise ValueError(f"{path}: unexpected size {len(blob)}")
    vals = struct.unpack("<11f", blob[:44])
    out = dict(zip(_SFS_FIELDS, vals))
    out["deltaTransform"] = np.frombuffer(blob[off_m:off_m + 64], dtype="<f4").reshape(4, 4).copy()
    out["lightingCoefficients"] = [float(v) for v in struct.unpack("<9f", blob[off_l:off_l + 36])]
    return out


def write_sfs_parameters(path, p):
    blob = struct.pack("<11f", *[p[k] for k in _SFS_FIELDS])
    blob += np.asarray(p.get("deltaTransform", np.eye(4)), dtype="<f4").tobytes()
    blob += struct.pack("<9f", *p["lightingCoefficients"]) + b"\0" * 16
    open(path, "wb").write(blob)


def read_constraints(path):
    toks = open(path).read().split()
    n = int(toks[0])
    v = [int(t) for t in toks[1:1 + 4 * n]]
    return [tuple(v[4 * i:4 * i + 4]) for i in range(n)]


def load_sfs_example(prefix, double=True):
    """Build the shape_from_shading problem from the reference's fixture set `<prefix>_targetIntensity.imagedump`, ...
    (examples/shape_from_shading/src/SFSSolverInput.h:22-66): binding order, the two uint8 edge masks stored as the two
    halves of one file, -inf depths clamped to -10000, weights and intrinsics from `<prefix>.SFSSolverParameters`."""
    ft = np.float64 if double else np.float32
    Im = read_imagedump(prefix + "_targetIntensity.imagedump", True)
    D = read_imagedump(prefix + "_targetDepth.imagedump", True)
    X = read_imagedump(prefix + "_initialUnknown.imagedump", True)
    edges = read_imagedump(prefix + "_maskEdgeMap.imagedump")
    H, W = D.shape
    flat = edges.reshape(-1)
    edgeR, edgeC = flat[:W * H].reshape(H, W).copy(), flat[W * H:2 * W * H].reshape(H, W).copy()
    p = read_sfs_parameters(prefix + ".SFSSolverParameters")
    f32 = lambda v: np.array(v, dtype=np.float32)
    params = [f32(p["weightFitting"]), f32(p["weightRegularizer"]), f32(p["weightShading"]), f32(p["fx"]), f32(p["fy"]), f32(p["ux"]), f32(p["uy"])]
    params += [f32(v) for v in p["lightingCoefficients"]]
    params += [X.astype(ft), D.astype(ft), Im.astype(ft), edgeR, edgeC]
    return wl.Problem("shape_from_shading", (W, H), params, (16,), double)


# ---- meshes and markers of the mesh examples (examples/data/*.off, *.ply, *.mrk) ---------------------------------------
def read_off(path):
    """Object File Format, triangles or polygons: returns (V float64 [n,3], F list of index lists)."""
    toks = open(path).read().split()
    assert toks[0] == "OFF", "not an OFF file"
    nv, nf = int(toks[1]), int(toks[2])
    V = np.array(toks[4:4 + 3 * nv], dtype=np.float64).reshape(nv, 3)
    F, i = [], 4 + 3 * nv
    for _ in range(nf):
        k = int(toks[i]); F.append([int(t) for t in toks[i + 1:i + 1 + k]]); i += 1 + k
    return V, F


def read_ply(path):
    """PLY with float x/y/z vertices and `list uchar int` faces, ascii or binary_little_endian (what the example meshes use)."""
    raw = open(path, "rb").read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    header = raw[:end].decode("ascii", "replace").split("\n")
    fmt = [l.split()[1] for l in header if l.startswith("format")][0]
    nv = int([l.split()[2] for l in header if l.startswith("element vertex")][0])
    nf = int([l.split()[2] for l in header if l.startswith("element face")][0])
    vprops = []
    section = None
    for l in header:
        if l.startswith("element"):
            section = l.split()[1]
        elif l.startswith("property") and section == "vertex":
            vprops.append(l.split()[1:])
    assert all(p[0] == "float" for p in vprops), "only float vertex properties are supported"
    names = [p[1] for p in vprops]
    body = raw[end:]
    if fmt == "ascii":
        toks = body.split()
        vals = np.array(toks[:nv * len(vprops)], dtype=np.float64).reshape(nv, len(vprops))
        F, i = [], nv * len(vprops)
        for _ in range(nf):
            k = int(toks[i]); F.append([int(t) for t in toks[i + 1:i + 1 + k]]); i += 1 + k
    else:
        assert fmt == "binary_little_endian"
        vals = np.frombuffer(body, dtype="<f4", count=nv * len(vprops)).reshape(nv, len(vprops)).astype(np.float64)
        F, off = [], 4 * nv * len(vprops)
        for _ in range(nf):
            k = body[off]; F.append(list(struct.unpack_from("<%di" % k, body, off + 1))); off += 1 + 4 * k
    V = np.stack([vals[:, names.index(c)] for c in "xyz"], 1)
    return V, F


def read_mrk(path):
    """Landmark file of the mesh examples (LandMarkSet.h): count, then `x y z radius vertexIndex` per line -> (indices, targets)."""
    toks = open(path).read().split()
    n = int(toks[0])
    rows = np.array(toks[1:1 + 5 * n], dtype=np.float64).reshape(n, 5)
    return rows[:, 4].astype(np.int64), rows[:, :3]


def mesh_half_edges(n_vertices, faces):
    """Directed half-edges (head -> neighbour) grouped by head vertex, one per vertex-vertex adjacency -- the graph the ARAP example
    builds from OpenMesh's vertex-vertex circulators (arap_mesh_deformation/src/CombinedSolver.h:105-129, OptGraph.h:64-76).
    Neighbours are listed in ascending index order (OpenMesh lists them in ring order; only the summation order differs)."""
    nb = [set() for _ in range(n_vertices)]
    for f in faces:
        for a, b in zip(f, f[1:] + f[:1]):
            nb[a].add(b); nb[b].add(a)
    heads = np.concatenate([np.full(len(s), v, dtype=np.int32) for v, s in enumerate(nb)]) if n_vertices else np.zeros(0, np.int32)
    tails = np.concatenate([np.array(sorted(s), dtype=np.int32) for s in nb]) if n_vertices else np.zeros(0, np.int32)
    return heads, tails


def arap_problem_from_mesh(V, faces, marker_idx, marker_pos, double=False, alpha=1.0):
    """The ARAP example's problem on a loaded mesh (CombinedSolver.h:62-103, 142-164): Offset = UrShape = vertices, Angle = 0,
    Constraints = -inf except the marker vertices, which are pulled a fraction `alpha` of the way to their marker positions
    (the example ramps alpha over its outer iterations); w_fit = 4, w_reg = 1."""
    ft = np.float64 if double else np.float32
    V = np.asarray(V, dtype=np.float64)
    N = len(V)
    cons = np.full((N, 3), -np.inf)
    for i, pos in zip(marker_idx, marker_pos):
        cons[int(i)] = (1 - alpha) * V[int(i)] + alpha * np.asarray(pos)
    heads, tails = mesh_half_edges(N, faces)
    return wl.Problem("arap_mesh_deformation", (N,),
                      [np.array(np.sqrt(np.float32(4.0)), dtype=np.float32), np.array(np.sqrt(np.float32(1.0)), dtype=np.float32),
                       V.astype(ft), np.zeros((N, 3), dtype=ft), V.astype(ft), cons.astype(ft), np.array(len(heads), dtype=np.int32), heads, tails],
                      (2, 3), double, {"n_edges": int(len(heads))})


# ---- PNG (the image examples' inputs: examples/data/*.png) -------------------------------------------------------------
def read_png(path):
    """Minimal PNG reader (zlib + the five scanline filters): 8-bit grey / grey-alpha / RGB / RGBA / palette, non-interlaced.
    Returns uint8 [H, W, channels] (palette images are expanded to RGB).  Enough for the reference's example images and masks."""
    import zlib
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG file"
    pos, idat, plte = 8, b"", None
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            W, H, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", data)
        elif typ == b"PLTE":
            plte = np.frombuffer(data, dtype=np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat += data
        elif typ == b"IEND":
            break
        pos += 12 + n
    assert depth == 8 and interlace == 0, "only 8-bit non-interlaced PNGs are supported"
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    stride = W * ch
    buf = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(H, stride + 1)
    out = np.zeros((H, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(H):
        f, line = int(buf[y, 0]), buf[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:                                   # Sub / Average / Paeth depend on the already reconstructed pixel to the left
            cur = np.zeros(stride, dtype=np.int32)
            for i in range(stride):
                a = cur[i - ch] if i >= ch else 0
                b = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if f == 1:
                    pred = a
                elif f == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    img = out.reshape(H, W, ch)
    if ctype == 3:
        img = plte[img[..., 0]]
    return img


def image_warping_problem_from_mask(mask_red, markers=wl.CAT512_MARKERS, downsample=1, double=False):
    """The image_warping example's inputs from its mask image (examples/image_warping/src/main.cpp:54-108, CombinedSolver.h:161-207):
    Mask = red channel of <image>_mask.png sampled every `downsample` pixels (a pixel is solved where it is 0), UrShape = Offset =
    pixel lattice, Angle = 0, Constraints = (-1,-1) except the markers (source -> target, divided by `downsample`) and ... every
    image-border pixel pinned to itself like the synthetic metric workload."""
    m = np.asarray(mask_red)[::downsample, ::downsample].astype(np.float64)
    H, W = m.shape
    P = wl.image_warping(W, H, double=double)
    ft = np.float64 if double else np.float32
    P.params[4] = m.astype(ft)
    cons = np.full((H, W, 2), -1.0)
    cons[0, :, 0] = np.arange(W); cons[0, :, 1] = 0; cons[-1, :, 0] = np.arange(W); cons[-1, :, 1] = H - 1
    cons[:, 0, 0] = 0; cons[:, 0, 1] = np.arange(H); cons[:, -1, 0] = W - 1; cons[:, -1, 1] = np.arange(H)
    for sx, sy, tx, ty in markers:
        x, y = sx // downsample, sy // downsample
        if 0 <= x < W and 0 <= y < H:
            cons[y, x] = (tx / downsample, ty / downsample)
    P.params[3] = cons.astype(ft)
    return P


def mesh_vertex_rings(n_vertices, faces):
    """Ordered one-ring of every vertex of a manifold triangle mesh (what OpenMesh's vertex-vertex circulator yields, up to the
    starting point and orientation): list of neighbour-index lists.  Interior vertices give a closed ring; at a boundary vertex the
    ring starts at the boundary edge that has no predecessor."""
    nxt = [dict() for _ in range(n_vertices)]          # around v: neighbour a is followed by neighbour b in triangle (v, a, b)
    for f in faces:
        assert len(f) == 3, "triangle meshes only"
        for i in range(3):
            v, a, b = f[i], f[(i + 1) % 3], f[(i + 2) % 3]
            nxt[v][a] = b
    rings = []
    for v in range(n_vertices):
        m = nxt[v]
        if not m:
            rings.append([]); continue
        followers = set(m.values())
        starts = [a for a in m if a not in followers]
        cur = min(starts) if starts else min(m)
        ring, seen = [], set()
        while cur is not None and cur not in seen:
            ring.append(cur); seen.add(cur)
            cur = m.get(cur)
        rings.append(ring)
    return rings
