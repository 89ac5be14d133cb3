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
