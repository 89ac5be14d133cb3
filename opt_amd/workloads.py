"""Synthetic problem instances for the energies this backend accelerates.

Host-side numpy only.  Every generator returns a `Problem` whose `params` list is ordered by the binding
index each array / scalar has in the energy's .t file (what a caller puts in Opt_ProblemSolve's void**).
The shapes mirror the reference example harnesses (cited per generator); there is no network, so image /
mesh payloads are procedural.
"""
from dataclasses import dataclass, field

import numpy as np

# examples/data/cat512.constraints (public-domain marker list: source x,y -> target x,y on a 512^2 image)
CAT512_MARKERS = [
    (30, 132, 59, 44), (229, 51, 157, 91), (430, 124, 379, 42), (281, 369, 326, 323), (197, 407, 163, 418),
    (64, 386, 26, 300), (311, 168, 253, 182), (89, 228, 56, 255), (92, 192, 84, 192),
]


@dataclass
class Problem:
    energy: str                 # .t file stem
    dims: tuple                 # Opt_ProblemPlan dimensions
    params: list                # host arrays / scalars by binding index
    unknown_slots: tuple        # which entries of `params` are unknowns (updated in place by a solve)
    double: bool = False
    meta: dict = field(default_factory=dict)

    def clone(self):
        return Problem(self.energy, self.dims, [np.array(p, copy=True) for p in self.params], self.unknown_slots,
                       self.double, dict(self.meta))


def image_warping(W, H=None, double=False, random_state=None, mask_fraction=0.0, perturb=0.0, jitter_urshape=0.0,
                  fit_fraction=0.0, w_fit_sqrt=None, w_reg_sqrt=None):
    """examples/image_warping/src/CombinedSolver.h:110-207 + main.cpp:98-108, constraint ramp alpha=1.

    random_state/mask_fraction/perturb > 0 give the randomised variant used by the parity tests
    (masked pixels, non-zero angles, perturbed offsets) so every branch of the energy is exercised.
    jitter_urshape > 0 moves the rest positions off the pixel lattice (the reference example always uses the
    lattice itself; the backend has a fast path for it and a general path for everything else).
    """
    H = H or W
    ft = np.float64 if double else np.float32
    xs, ys = np.meshgrid(np.arange(W, dtype=ft), np.arange(H, dtype=ft))
    urshape = np.stack([xs, ys], axis=-1).astype(ft)                  # (H,W,2), x fastest
    offset = urshape.copy()
    angle = np.zeros((H, W), dtype=ft)
    mask = np.zeros((H, W), dtype=ft)
    constraints = np.full((H, W, 2), -1.0, dtype=ft)
    # border pixels pinned to themselves (main.cpp:98-108)
    constraints[0, :, :] = urshape[0, :, :]
    constraints[-1, :, :] = urshape[-1, :, :]
    constraints[:, 0, :] = urshape[:, 0, :]
    constraints[:, -1, :] = urshape[:, -1, :]
    sx, sy = W / 512.0, H / 512.0
    for (x0, y0, x1, y1) in CAT512_MARKERS:
        x, y = int(x0 * sx), int(y0 * sy)
        if 0 <= x < W and 0 <= y < H:
            constraints[y, x] = (int(x1 * sx), int(y1 * sy))
    if random_state is not None:
        rng = np.random.default_rng(random_state)
        if jitter_urshape > 0:
            urshape += (jitter_urshape * rng.standard_normal((H, W, 2))).astype(ft)
            offset = urshape.copy()
        if mask_fraction > 0:
            mask[rng.random((H, W)) < mask_fraction] = 255.0
        if perturb > 0:
            offset += (perturb * rng.standard_normal((H, W, 2))).astype(ft)
            angle += (0.3 * perturb * rng.standard_normal((H, W))).astype(ft)
            extra = rng.random((H, W)) < 0.05
            constraints[extra] = (urshape[extra] + 3.0 * rng.standard_normal((int(extra.sum()), 2))).astype(ft)
            constraints[extra] = np.abs(constraints[extra])
    if fit_fraction > 0:
        rng2 = np.random.default_rng(1234 if random_state is None else random_state + 1)
        extra = rng2.random((H, W)) < fit_fraction
        constraints[extra] = np.abs(urshape[extra] + 2.0 * rng2.standard_normal((int(extra.sum()), 2))).astype(ft)
    w_fit = np.array(np.sqrt(np.float32(100.0)) if w_fit_sqrt is None else w_fit_sqrt, dtype=np.float32)     # CombinedSolver.h:126-130
    w_reg = np.array(np.sqrt(np.float32(0.01)) if w_reg_sqrt is None else w_reg_sqrt, dtype=np.float32)
    return Problem("image_warping", (W, H), [offset, angle, urshape, constraints, mask, w_fit, w_reg], (0, 1), double)


def image_warping_rows(W, H, lo, hi, double=False):
    """Rows [lo, hi) of image_warping(W, H) (the deterministic variant: no random_state) without building the whole image -- what one rank of
    a multi-GPU slab job needs (its rows + ghost rows; opt_amd/slab.py).  Rows outside [0, H) are zero-filled with Mask = 255, exactly what
    slab.split_problem makes of the global problem.  dims = (W, hi - lo)."""
    ft = np.float64 if double else np.float32
    n = hi - lo
    glo, ghi = max(lo, 0), min(hi, H)
    inside = slice(glo - lo, ghi - lo)
    ys_g = np.arange(glo, ghi, dtype=ft)
    xs, ys = np.meshgrid(np.arange(W, dtype=ft), ys_g)
    urshape = np.zeros((n, W, 2), dtype=ft)
    urshape[inside] = np.stack([xs, ys], axis=-1)
    offset = urshape.copy()
    angle = np.zeros((n, W), dtype=ft)
    mask = np.zeros((n, W), dtype=ft)
    if lo < 0:
        mask[:-lo] = 255
    if hi > H:
        mask[n - (hi - H):] = 255
    constraints = np.zeros((n, W, 2), dtype=ft)
    constraints[inside] = -1.0
    for gy in (0, H - 1):                                  # border rows pinned to themselves
        if glo <= gy < ghi:
            constraints[gy - lo] = urshape[gy - lo]
    constraints[inside, 0, :] = urshape[inside, 0, :]
    constraints[inside, -1, :] = urshape[inside, -1, :]
    sx, sy = W / 512.0, H / 512.0
    for (x0, y0, x1, y1) in CAT512_MARKERS:
        x, y = int(x0 * sx), int(y0 * sy)
        if 0 <= x < W and glo <= y < ghi:
            constraints[y - lo, x] = (int(x1 * sx), int(y1 * sy))
    w_fit = np.array(np.sqrt(np.float32(100.0)), dtype=np.float32)
    w_reg = np.array(np.sqrt(np.float32(0.01)), dtype=np.float32)
    return Problem("image_warping", (W, n), [offset, angle, urshape, constraints, mask, w_fit, w_reg], (0, 1), double)


def poisson_image_editing(W, H=None, double=False, seed=0):
    """examples/poisson_image_editing/src/CombinedSolver.h:29-50, 66-90: float4 base image X, inserted image T,
    mask M = 0 inside the pasted region (solved), 255 outside (kept)."""
    H = H or W
    ft = np.float64 if double else np.float32
    rng = np.random.default_rng(seed)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    base = np.stack([128 + 100 * np.sin(xs / 17.0), 128 + 100 * np.cos(ys / 23.0), 0.5 * (xs + ys) % 255, np.full_like(xs, 255.0, dtype=float)], -1)
    ins = rng.uniform(0, 255, size=(H, W, 4))
    ins[..., 3] = 255.0
    M = np.full((H, W), 255.0)
    M[H // 4: H // 4 + H // 2, W // 4: W // 4 + W // 2] = 0.0
    return Problem("poisson_image_editing", (W, H), [base.astype(ft), ins.astype(ft), M.astype(ft)], (0,), double)


def laplacian(W, H=None, seed=0):
    """tests/minimal/main.cpp:44-56: random target in [0,1], unknown initialised to the target."""
    H = H or W
    rng = np.random.default_rng(seed)
    A = rng.random((H, W)).astype(np.float32)
    return Problem("laplacian", (W, H), [A.copy(), A], (0,), False)


def curve_fitting(n=512, double=True):
    """tests/minimal_graph_only/main.cpp:43-90: y = a cos(bx) + b sin(ax), (a,b) = (100,102), start (99.7,101.6)."""
    ft = np.float64 if double else np.float32
    a, b = 100.0, 102.0
    x = (np.arange(n, dtype=np.float32).astype(np.float64) * 2.0 * 3.141592653589 / n)
    y = a * np.cos(b * x) + b * np.sin(a * x)
    data = np.stack([x, y], -1).astype(ft)
    unknown = np.array([[np.float32(99.7), np.float32(101.6)]], dtype=ft)
    n_edges = np.array(n, dtype=np.int32)
    end_nodes = np.arange(n, dtype=np.int32)      # "d": data index
    start_nodes = np.zeros(n, dtype=np.int32)     # "p": parameter index
    return Problem("curveFitting", (n, 1), [unknown, data, n_edges, end_nodes, start_nodes], (0,), double,
                   {"goal": (a, b)})


def grid_mesh_edges(nx, ny):
    """Directed half-edges of a regular triangulated grid, grouped by head vertex exactly like
    createGraphFromNeighborLists (examples/shared/OptGraph.h:64-76)."""
    idx = np.arange(nx * ny).reshape(ny, nx)
    nbr = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1)]      # 6-valence interior
    heads, tails = [], []
    ys, xs = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    cols = []
    for dx, dy in nbr:
        tx, ty = xs + dx, ys + dy
        ok = (tx >= 0) & (tx < nx) & (ty >= 0) & (ty < ny)
        t = np.where(ok, idx[np.clip(ty, 0, ny - 1), np.clip(tx, 0, nx - 1)], -1)
        cols.append(t.reshape(-1))
    T = np.stack(cols, 1)                                            # (N,6), -1 where absent
    h = np.repeat(np.arange(nx * ny), 6).reshape(-1, 6)
    keep = T >= 0
    heads = h[keep].astype(np.int32)
    tails = T[keep].astype(np.int32)
    return heads, tails


def arap_mesh_deformation(nx, ny=None, double=False, seed=0, perturb=0.0):
    """examples/arap_mesh_deformation/src/CombinedSolver.h:62-164 on a procedural grid mesh (the reference's
    mesh blobs are not shipped): UrShape = Offset = rest pose, Angle = 0, Constraints = -inf except handles."""
    ny = ny or nx
    ft = np.float64 if double else np.float32
    N = nx * ny
    ys, xs = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    rest = np.stack([xs.reshape(-1) * 0.01, ys.reshape(-1) * 0.01, 0.02 * np.sin(xs.reshape(-1) * 0.1) * np.cos(ys.reshape(-1) * 0.13)], -1).astype(ft)
    offset = rest.copy()
    angle = np.zeros((N, 3), dtype=ft)
    cons = np.full((N, 3), -np.inf, dtype=ft)
    # handles: pin the left column, displace the right column (like the .mrk marker files)
    left = (xs.reshape(-1) == 0)
    right = (xs.reshape(-1) == nx - 1)
    cons[left] = rest[left]
    cons[right] = rest[right] + np.array([0.0, 0.05 * ny * 0.01, 0.1 * nx * 0.01], dtype=ft)
    if perturb > 0:
        rng = np.random.default_rng(seed)
        offset += (perturb * rng.standard_normal((N, 3))).astype(ft)
        angle += (perturb * 5 * rng.standard_normal((N, 3))).astype(ft)
    heads, tails = grid_mesh_edges(nx, ny)
    w_fit = np.array(np.sqrt(np.float32(4.0)), dtype=np.float32)      # main.cpp:103-104
    w_reg = np.array(np.sqrt(np.float32(1.0)), dtype=np.float32)
    n_edges = np.array(len(heads), dtype=np.int32)
    return Problem("arap_mesh_deformation", (N,), [w_fit, w_reg, offset, angle, rest, cons, n_edges, heads, tails], (2, 3), double,
                   {"n_edges": int(len(heads))})


# decoded from examples/data/shape_from_shading/default.SFSSolverParameters (struct layout TerraSolverParameters.h:7-44)
SFS_FIXTURE = dict(w_p=100.0, w_s=100.0, w_g=1.0, fx=574.0529, fy=574.0528, ux=320.0, uy=240.0,
                   L=[0.6908, 0.0446, 0.0181, -0.1773, -0.0407, 0.1447, 0.0239, -0.2466, 0.0058])


def _sfs_render(depth, fx, fy, ux, uy, L):
    """Shading B of shape_from_shading.t:32-54 evaluated on a depth map (numpy, float64)."""
    H, W = depth.shape
    j, i = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    d1 = depth
    d0 = np.roll(depth, 1, axis=1)     # X(-1,0)
    d2 = np.roll(depth, 1, axis=0)     # X(0,-1)
    nx = d2 * (d1 - d0) / fy
    ny = d0 * (d1 - d2) / fx
    nz = nx * (ux - i) / fx + ny * (uy - j) / fy - d0 * d2 / (fx * fy)
    sq = nx * nx + ny * ny + nz * nz
    inv = np.where(sq > 0, 1.0 / np.sqrt(np.where(sq > 0, sq, 1.0)), 1.0)
    nx, ny, nz = nx * inv, ny * inv, nz * inv
    return (L[0] + L[1] * ny + L[2] * nz + L[3] * nx + L[4] * nx * ny + L[5] * ny * nz +
            L[6] * (-nx * nx - ny * ny + 2 * nz * nz) + L[7] * nz * nx + L[8] * (nx * nx - ny * ny))


def shape_from_shading(W, H=None, double=True, seed=0, holes=False, noise=1e-3):
    """examples/shape_from_shading/src/SFSSolverInput.h:22-47 binding order; synthetic smooth depth,
    intensity rendered from it with the fixture's SH lighting, intrinsics scaled from the 640x480 fixture."""
    H = H or W
    ft = np.float64 if double else np.float32
    rng = np.random.default_rng(seed)
    j, i = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    depth = 0.45 + 0.05 * np.sin(i / W * 6.0) * np.cos(j / H * 5.0)
    fx = SFS_FIXTURE["fx"] * W / 640.0
    fy = SFS_FIXTURE["fy"] * W / 640.0
    ux, uy = W / 2.0, H / 2.0
    L = SFS_FIXTURE["L"]
    shading = _sfs_render(depth, fx, fy, ux, uy, L)
    # target intensity: rendered shading of a slightly different (detail-carrying) surface
    detail = depth + 0.002 * np.sin(i * 0.7) * np.sin(j * 0.9)
    Im = _sfs_render(detail, fx, fy, ux, uy, L)
    D_i = depth.copy()
    if holes:
        hole = rng.random((H, W)) < 0.03
        D_i[hole] = -10000.0          # SimpleBuffer.cpp:30-41 clamps -inf depths to a large negative
    X = depth + noise * rng.standard_normal((H, W))
    X = np.where(D_i > 0, X, D_i)
    edgeR = np.ones((H, W), dtype=np.uint8)
    edgeC = np.ones((H, W), dtype=np.uint8)
    if holes:
        edgeR[rng.random((H, W)) < 0.05] = 0
        edgeC[rng.random((H, W)) < 0.05] = 0
    f32 = lambda v: np.array(v, dtype=np.float32)
    params = [f32(SFS_FIXTURE["w_p"]), f32(SFS_FIXTURE["w_s"]), f32(SFS_FIXTURE["w_g"]), f32(fx), f32(fy), f32(ux), f32(uy)]
    params += [f32(v) for v in L]
    params += [X.astype(ft), D_i.astype(ft), Im.astype(ft), edgeR, edgeC]
    return Problem("shape_from_shading", (W, H), params, (16,), double)


def _smooth_image(W, H, rng, octaves=4):
    """A band-limited random image in [0, 1] (sum of a few random sinusoids): stands in for a blurred photograph."""
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    im = np.zeros((H, W))
    for o in range(octaves):
        fx, fy = rng.uniform(0.02, 0.12, 2) * (1.6 ** o)
        im += rng.uniform(0.3, 1.0) / (1.5 ** o) * np.sin(fx * xs + rng.uniform(0, 6.3)) * np.cos(fy * ys + rng.uniform(0, 6.3))
    im -= im.min()
    return im / max(im.max(), 1e-12)


def optical_flow(W, H=None, double=False, seed=0, init_flow=0.0):
    """examples/optical_flow/src/CombinedSolver.h:72-83, 107-171: source / target frames, the target's derivative images
    (3x3 Prewitt / 8, zero border), flow initialised to 0 (init_flow > 0: seeded random flow, so tests sample off-lattice);
    w_fit = sqrt(10), w_reg = sqrt(0.1)."""
    H = H or W
    ft = np.float64 if double else np.float32
    rng = np.random.default_rng(seed)
    target = _smooth_image(W + 8, H + 8, rng)
    source = target[3:3 + H, 5:5 + W].copy()                 # the source frame is the target moved by (+1, -1) px
    target = target[4:4 + H, 4:4 + W].copy()
    du = np.zeros((H, W)); dv = np.zeros((H, W))
    du[1:-1, 1:-1] = (-target[:-2, :-2] - target[1:-1, :-2] - target[2:, :-2] + target[:-2, 2:] + target[1:-1, 2:] + target[2:, 2:]) / 8.0
    dv[1:-1, 1:-1] = (-target[:-2, :-2] - target[:-2, 1:-1] - target[:-2, 2:] + target[2:, :-2] + target[2:, 1:-1] + target[2:, 2:]) / 8.0
    X = rng.uniform(-init_flow, init_flow, size=(H, W, 2)) if init_flow > 0 else np.zeros((H, W, 2))
    return Problem("optical_flow", (W, H), [np.float32(np.sqrt(10.0)), np.float32(np.sqrt(0.1)), X.astype(ft), source.astype(ft), target.astype(ft),
                                            du.astype(ft), dv.astype(ft)], (2,), double)


def intrinsic_image_decomposition(W, H=None, double=False, seed=0):
    """examples/intrinsic_image_decomposition/src/CombinedSolver.h:28-45, 60-90: log input image i (float3), unknown log-albedo r
    initialised to the input, unknown log-shading s initialised to 0 (here + small seeded noise so gradients are generic);
    weights 500 / 1000 / 10000 (sqrt taken by the caller), p = 0.8."""
    H = H or W
    ft = np.float64 if double else np.float32
    rng = np.random.default_rng(seed)
    albedo = np.stack([np.round(3 * _smooth_image(W, H, rng)) / 3 for _ in range(3)], -1) * 0.8 + 0.1      # piecewise constant
    shading = 0.4 + 0.6 * _smooth_image(W, H, rng, octaves=2)
    i = np.log(albedo * shading[..., None] + 1e-3)
    r = i + rng.normal(0, 0.02, size=i.shape)
    s = rng.normal(0, 0.02, size=(H, W))
    return Problem("intrinsic_image_decomposition", (W, H),
                   [np.float32(np.sqrt(500.0)), np.float32(np.sqrt(1000.0)), np.float32(np.sqrt(10000.0)), ft(0.8), r.astype(ft), i.astype(ft), s.astype(ft)],
                   (4, 6), double)


def volumetric_mesh_deformation(W, H=None, D=None, double=False, seed=0, perturb=0.0):
    """examples/volumetric_mesh_deformation/src/CombinedSolver.h:40-60, 95-160: regular lattice, Offset = UrShape = lattice positions,
    Angle = 0, constraints = -inf except the bottom layer (pinned) and the top layer (moved and twisted); w_fit = 1, w_reg = 0.05."""
    H = H or W; D = D or W
    ft = np.float64 if double else np.float32
    rng = np.random.default_rng(seed)
    zs, ys, xs = np.meshgrid(np.arange(D, dtype=np.float64), np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    ur = np.stack([xs, ys, zs], -1) * 0.1
    cons = np.full(ur.shape, -np.inf)
    cons[0] = ur[0]
    th = 0.4
    top = ur[-1].copy()
    cx, cy = top[..., 0].mean(), top[..., 1].mean()
    tx, ty = top[..., 0] - cx, top[..., 1] - cy
    cons[-1] = np.stack([cx + np.cos(th) * tx - np.sin(th) * ty + 0.05, cy + np.sin(th) * tx + np.cos(th) * ty, top[..., 2] + 0.1], -1)
    off = ur + (rng.normal(0, perturb, size=ur.shape) if perturb > 0 else 0)
    ang = rng.normal(0, perturb, size=ur.shape) if perturb > 0 else np.zeros(ur.shape)
    return Problem("volumetric_mesh_deformation", (W, H, D), [off.astype(ft), ang.astype(ft), ur.astype(ft), cons.astype(ft), np.float32(1.0), np.float32(np.sqrt(0.05))],
                   (0, 1), double)


def grid_surface_mesh(nx, ny, seed=0, bump=0.3):
    """A triangulated height-field patch: (vertices float64 [nx*ny, 3], triangle list).  Stands in for the example meshes."""
    rng = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.arange(ny, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing="ij")
    z = bump * np.sin(0.9 * xs + rng.uniform(0, 3)) * np.cos(0.7 * ys + rng.uniform(0, 3))
    V = np.stack([xs.reshape(-1), ys.reshape(-1), z.reshape(-1)], -1)
    F = []
    for y in range(ny - 1):
        for x in range(nx - 1):
            a, b, c, d = y * nx + x, y * nx + x + 1, (y + 1) * nx + x, (y + 1) * nx + x + 1
            F.append([a, b, d]); F.append([a, d, c])
    return V, F


def torus_mesh(nx, ny, seed=0, R=3.0, r=1.0):
    """A closed triangulated torus (every vertex has a full ring of 6): (vertices float64 [nx*ny, 3], triangle list).  Stands in for
    the closed example meshes of the smoothing example."""
    rng = np.random.default_rng(seed)
    js, is_ = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    u, v = 2 * np.pi * is_ / nx, 2 * np.pi * js / ny
    rr = r * (1 + 0.15 * np.sin(3 * u + rng.uniform(0, 3)))
    V = np.stack([(R + rr * np.cos(v)) * np.cos(u), (R + rr * np.cos(v)) * np.sin(u), rr * np.sin(v)], -1).reshape(-1, 3)
    F = []
    for y in range(ny):
        for x in range(nx):
            a, b = y * nx + x, y * nx + (x + 1) % nx
            c, d = ((y + 1) % ny) * nx + x, ((y + 1) % ny) * nx + (x + 1) % nx
            F.append([a, b, d]); F.append([a, d, c])
    return V, F


def _half_edges_from_rings(rings):
    heads = np.concatenate([np.full(len(r), v, dtype=np.int32) for v, r in enumerate(rings)])
    tails = np.concatenate([np.array(r, dtype=np.int32) for r in rings])
    return heads, tails


def cotangent_mesh_smoothing(nx, ny=None, double=False, seed=0, noise=0.05):
    """examples/cotangent_mesh_smoothing/src/CombinedSolver.h:16-44, 68-113: X = A = the (noisy) input vertices; one hyperedge per
    (vertex, ring neighbour): v0 = vertex, v1 = neighbour, v2 / v3 = previous / next neighbour in the ring (cyclic);
    w_fit = 1, w_reg = 0.5 (main.cpp:34-35)."""
    from . import io
    ny = ny or nx
    ft = np.float64 if double else np.float32
    V, F = torus_mesh(nx, ny, seed)          # closed, like the example's meshes: no hyperedge lists a vertex twice
    V = V + np.random.default_rng(seed + 1).normal(0, noise, size=V.shape)
    rings = io.mesh_vertex_rings(len(V), F)
    v0, v1, v2, v3 = [], [], [], []
    for v, r in enumerate(rings):
        n = len(r)
        for i in range(n):
            v0.append(v); v1.append(r[i]); v2.append(r[(i + n - 1) % n]); v3.append(r[(i + 1) % n])
    idx = [np.array(a, dtype=np.int32) for a in (v0, v1, v2, v3)]
    return Problem("cotangent_mesh_smoothing", (len(V),),
                   [np.float32(np.sqrt(1.0)), np.float32(np.sqrt(0.5)), V.astype(ft), V.astype(ft), np.array(len(v0), dtype=np.int32)] + idx, (2,), double,
                   {"n_edges": len(v0)})


def embedded_mesh_deformation(nx, ny=None, double=False, seed=0, perturb=0.0):
    """examples/embedded_mesh_deformation/src/CombinedSolver.h:33-56, 95-160: Offset = UrShape = node positions, RotMatrix = identity,
    Constraints = -inf except handle nodes (one edge column pinned, the opposite one lifted); weights 3 / 12 / 5."""
    from . import io
    ny = ny or nx
    ft = np.float64 if double else np.float32
    V, F = grid_surface_mesh(nx, ny, seed)
    rng = np.random.default_rng(seed + 2)
    heads, tails = _half_edges_from_rings(io.mesh_vertex_rings(len(V), F))
    off = V + (rng.normal(0, perturb, size=V.shape) if perturb > 0 else 0)
    rot = np.tile(np.eye(3).reshape(-1), (len(V), 1)) + (rng.normal(0, perturb, size=(len(V), 9)) if perturb > 0 else 0)
    cons = np.full(V.shape, -np.inf)
    left, right = np.arange(ny) * nx, np.arange(ny) * nx + nx - 1
    cons[left] = V[left]; cons[right] = V[right] + np.array([0.0, 0.3, 0.8])
    return Problem("embedded_mesh_deformation", (len(V),),
                   [np.float32(np.sqrt(3.0)), np.float32(np.sqrt(12.0)), np.float32(np.sqrt(5.0)), off.astype(ft), rot.astype(ft), V.astype(ft), cons.astype(ft),
                    np.array(len(heads), dtype=np.int32), heads, tails], (3, 4), double, {"n_edges": int(len(heads))})


def robust_nonrigid_alignment(nx, ny=None, double=False, seed=0, perturb=0.0):
    """examples/robust_nonrigid_alignment/src/CombinedSolver.h:150-180: Offset = UrShape = source vertices, Angle = 0, RobustWeights = 1,
    Constraints / ConstraintNormals = the corresponding point and normal on the target surface where a correspondence exists
    (-inf otherwise); w_fit = 10, w_reg = 64 (its starting value)."""
    from . import io
    ny = ny or nx
    ft = np.float64 if double else np.float32
    V, F = grid_surface_mesh(nx, ny, seed)
    rng = np.random.default_rng(seed + 3)
    heads, tails = _half_edges_from_rings(io.mesh_vertex_rings(len(V), F))
    target = V + np.stack([0.05 * np.sin(V[:, 1]), 0.04 * np.cos(V[:, 0]), 0.3 + 0.1 * np.sin(0.5 * V[:, 0])], -1)
    nrm = np.stack([-0.05 * np.cos(0.5 * V[:, 0]), 0.02 * np.sin(V[:, 1]), np.ones(len(V))], -1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    has = rng.random(len(V)) < 0.7
    cons = np.where(has[:, None], target, -np.inf)
    off = V + (rng.normal(0, perturb, size=V.shape) if perturb > 0 else 0)
    ang = rng.normal(0, perturb, size=V.shape) if perturb > 0 else np.zeros(V.shape)
    rw = 1.0 + (rng.normal(0, perturb, size=len(V)) if perturb > 0 else 0) * np.ones(len(V))
    return Problem("robust_nonrigid_alignment", (len(V),),
                   [np.float32(np.sqrt(10.0)), np.float32(np.sqrt(64.0)), off.astype(ft), ang.astype(ft), rw.astype(ft), V.astype(ft), cons.astype(ft), nrm.astype(ft),
                    np.array(len(heads), dtype=np.int32), heads, tails], (2, 3, 4), double, {"n_edges": int(len(heads))})
