"""CPU tests of the oracle itself (-m "not gpu").

The reference ships no golden vectors, so the oracle is pinned by
  * the reference's one known-answer test, tests/minimal_graph_only (curve fit -> (100,102)),
  * identities any correct restatement of Opt's generator must satisfy: J^T F equals the finite-difference
    gradient of the cost, J^T J is symmetric PSD with diag equal to the preconditioner diagonal, and
  * the committed golden fixtures (tests/golden), which freeze today's oracle output against regressions.
"""
import numpy as np
import pytest

from opt_amd import workloads as wl
from helpers import flat_unknowns, oracle_solver, rel_err


def test_known_answer_curve_fitting(oracle_lib):
    # reference tests/minimal_graph_only/main.cpp:43-61, 88-90: double precision, GN defaults (10 x 10)
    P = wl.curve_fitting()
    s = oracle_solver(oracle_lib, P)
    s.solve(P.params)
    a, b = P.params[0].reshape(-1)
    assert abs(a - 100.0) < 1e-9 and abs(b - 102.0) < 1e-9
    assert s.cost() < 1e-12
    hist = s.cost_history()
    assert hist[0] > 1e6 and np.all(np.diff(hist[:6]) < 0)


def _fd_gradient(s, P, h):
    g = []
    for slot in P.unknown_slots:
        X = P.params[slot].reshape(-1)
        for i in range(X.size):
            old = X[i]
            X[i] = old + h; cp = s.eval_cost(P.params)
            X[i] = old - h; cm = s.eval_cost(P.params)
            X[i] = old
            g.append((cp - cm) / (2 * h))
    return np.array(g)


def _active_mask(P):
    if P.energy == "image_warping":
        m = np.asarray(P.params[4]).reshape(-1) == 0
        return np.concatenate([np.repeat(m, 2), m])
    if P.energy == "poisson_image_editing":
        return np.repeat(np.asarray(P.params[2]).reshape(-1) == 0, 4)
    if P.energy == "shape_from_shading":
        return np.asarray(P.params[17]).reshape(-1) > 0
    return np.ones(flat_unknowns(P).size, dtype=bool)


CASES = {
    "image_warping": lambda: wl.image_warping(9, 7, double=True, random_state=1, mask_fraction=0.1, perturb=0.5),
    "poisson_image_editing": lambda: wl.poisson_image_editing(8, 6, double=True, seed=3),
    "arap_mesh_deformation": lambda: wl.arap_mesh_deformation(5, 4, double=True, seed=2, perturb=0.02),
    "curveFitting": lambda: wl.curve_fitting(16, double=True),
    "shape_from_shading": lambda: wl.shape_from_shading(12, 10, double=True, seed=4, holes=True, noise=2e-3),
    "optical_flow": lambda: wl.optical_flow(10, 8, double=True, seed=1, init_flow=0.7),
    "intrinsic_image_decomposition": lambda: wl.intrinsic_image_decomposition(8, 7, double=True, seed=2),
    "volumetric_mesh_deformation": lambda: wl.volumetric_mesh_deformation(4, 3, 3, double=True, seed=3, perturb=0.05),
    "cotangent_mesh_smoothing": lambda: wl.cotangent_mesh_smoothing(5, 4, double=True, seed=1),
    "embedded_mesh_deformation": lambda: wl.embedded_mesh_deformation(5, 4, double=True, seed=2, perturb=0.05),
    "robust_nonrigid_alignment": lambda: wl.robust_nonrigid_alignment(5, 4, double=True, seed=3, perturb=0.05),
}


# (optical_flow takes no part: the partials of its sample operator are the supplied derivative images (o.t:2494-2498), not d(bilinear)/dx -- J^T F is not the gradient of its cost)
@pytest.mark.parametrize("name", sorted(n for n in CASES if n != "optical_flow"))
def test_jtf_is_gradient_of_cost(oracle_lib, name):
    """F^ = J^T F must be d(cost)/dx on non-excluded rows.  For energies whose excluded centres carry
    residuals that touch active unknowns (poisson), the cost drops those rows while J^T F keeps them
    (o.t:2045-2064 has no exclude test) -- so the identity is checked with no pixel excluded there."""
    P = CASES[name]()
    if name == "poisson_image_editing":
        P.params[2][...] = 0
    if name == "intrinsic_image_decomposition":
        P.params[3] = np.float64(2.0)      # p = 2: the re-weighting factor is 1; for p != 2 it is a constant of the linearisation, not of the cost
    s = oracle_solver(oracle_lib, P)
    f, d = s.eval_jtf(P.params)
    g = _fd_gradient(s, P, 1e-7 if name == "shape_from_shading" else 1e-6)
    act = _active_mask(P)
    assert rel_err(f[act], g[act]) < (1e-5 if name == "shape_from_shading" else 1e-6)
    assert np.all(d >= 0)


@pytest.mark.parametrize("name", sorted(CASES))
def test_jtj_symmetric_psd_with_matching_diagonal(oracle_lib, name):
    P = CASES[name]()
    s = oracle_solver(oracle_lib, P)
    n = s.n
    act = _active_mask(P)
    rng = np.random.default_rng(0)
    a = rng.standard_normal(n) * act
    b = rng.standard_normal(n) * act
    Aa, Ab = s.apply_jtj(P.params, a), s.apply_jtj(P.params, b)
    assert abs(a @ Ab - b @ Aa) <= 1e-10 * (abs(a @ Ab) + 1)
    assert a @ Aa >= 0
    _, d = s.eval_jtf(P.params)
    idx = np.flatnonzero(act)[:: max(1, act.sum() // 25)]
    for i in idx:
        e = np.zeros(n); e[i] = 1
        assert abs(s.apply_jtj(P.params, e)[i] - d[i]) <= 1e-10 * (abs(d[i]) + 1)


def test_poisson_excluded_neighbours_still_contribute(oracle_lib):
    """SURVEY 8a 'exclude' row: residuals centred on excluded pixels still feed their active neighbours'
    J^T F / J^T J, so each in-bounds edge counts twice in diag(J^T J) regardless of the mask."""
    P = wl.poisson_image_editing(8, 8, double=True)
    s = oracle_solver(oracle_lib, P)
    _, d = s.eval_jtf(P.params)
    d = d.reshape(8, 8, 4)
    assert np.all(d[3, 3] == 8.0) and np.all(d[0, 0] == 4.0) and np.all(d[0, 3] == 6.0)


def test_gn_reduces_cost_image_warping(oracle_lib):
    P = wl.image_warping(24, 20, double=False)
    s = oracle_solver(oracle_lib, P, nIterations=3, lIterations=20)
    s.solve(P.params)
    h = s.cost_history()
    assert len(h) == 4 and h[-1] < h[0] * 0.5
    tr = s.trace()
    assert tr.shape == (60, 6) and np.all(tr[:, 3] > 0)


def test_lm_accepts_and_updates_radius(oracle_lib):
    P = wl.image_warping(16, 16, double=True)
    s = oracle_solver(oracle_lib, P, "LMGPU", nIterations=4, lIterations=30)
    s.init(P.params)
    c0 = s.cost()
    while s.step(P.params):
        pass
    assert s.cost() < c0
    assert s.trust_region_radius() != 1e4


def test_unknown_energy_or_kind_rejected(oracle_lib):
    with pytest.raises(ValueError):
        oracle_lib.OracleSolver("no_such_energy", "gaussNewtonGPU", False, (4, 4))
    with pytest.raises(ValueError):
        oracle_lib.OracleSolver("image_warping", "gradientDescentCPU", False, (4, 4))   # o.t:122


def _laplacian_system(A):
    """Residual matrix of tests/minimal/laplacian.t written out by hand: rows 0.2 (X - A), X(0,0) - X(1,0), X(0,0) - X(0,1);
    a row whose stencil leaves the image is dropped (it is identically 0, o.t:1930-1933)."""
    import scipy.sparse as sp
    H, W = A.shape
    idx = np.arange(H * W).reshape(H, W)
    rows, cols, vals, rhs = [], [], [], []
    def add(entries, b):
        r = len(rhs)
        for c, v in entries:
            rows.append(r); cols.append(c); vals.append(v)
        rhs.append(b)
    for y in range(H):
        for x in range(W):
            add([(idx[y, x], 0.2)], 0.2 * A[y, x])
            if x + 1 < W:
                add([(idx[y, x], 1.0), (idx[y, x + 1], -1.0)], 0.0)
            if y + 1 < H:
                add([(idx[y, x], 1.0), (idx[y + 1, x], -1.0)], 0.0)
    return sp.csr_matrix((vals, (rows, cols)), shape=(len(rhs), H * W)), np.array(rhs)


def test_linear_energy_minimiser_matches_independent_least_squares(oracle_lib):
    """Known answer independent of the oracle's own machinery: laplacian.t is linear, so Gauss-Newton with enough
    PCG iterations must land on the least-squares solution of the residual system written out by hand and solved
    with scipy.  Pins cost scaling (1/2 sum r^2), boundary handling and the CG loop together."""
    import scipy.sparse.linalg as spla
    P = wl.laplacian(21, 17, seed=5)
    A = P.params[1].astype(np.float64)
    J, b = _laplacian_system(A)
    x_ls = spla.lsqr(J, b, atol=1e-14, btol=1e-14, iter_lim=20000)[0]
    cost_ls = 0.5 * np.sum((J @ x_ls - b) ** 2)
    s = oracle_solver(oracle_lib, P, nIterations=3, lIterations=400)
    s.solve(P.params)
    assert rel_err(P.params[0].reshape(-1), x_ls) < 2e-4            # float solver vs double least squares
    assert abs(s.cost() - cost_ls) <= 2e-5 * cost_ls


def test_poisson_minimiser_matches_independent_least_squares(oracle_lib):
    """Same for poisson_image_editing.t (double): unknowns = pixels with M == 0; every in-bounds directed edge (c, n) whose
    CENTRE c is unmasked is a residual row (X_c - X_n) - (T_c - T_n) in the cost; rows centred on masked pixels do not count
    towards the cost but still pull on their unmasked neighbour through J^T F -- so the stationary point GN
    reaches is that of the row set {centre unmasked} + {centre masked, neighbour unmasked}."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    P = wl.poisson_image_editing(14, 11, double=True, seed=2)
    X0, T, M = P.params[0].copy(), P.params[1], P.params[2]
    H, W = M.shape
    free = M == 0
    col = -np.ones((H, W), dtype=int); col[free] = np.arange(free.sum())
    rows, cols, vals, rhs = [], [], [], []
    for y in range(H):
        for x in range(W):
            for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                nx, ny = x + dx, y + dy
                if not (0 <= nx < W and 0 <= ny < H):
                    continue
                if not (free[y, x] or free[ny, nx]):
                    continue
                for k in range(4):
                    r = len(rhs)
                    bk = T[y, x, k] - T[ny, nx, k]
                    if free[y, x]:
                        rows.append(r); cols.append(4 * col[y, x] + k); vals.append(1.0)
                    else:
                        bk -= X0[y, x, k]
                    if free[ny, nx]:
                        rows.append(r); cols.append(4 * col[ny, nx] + k); vals.append(-1.0)
                    else:
                        bk += X0[ny, nx, k]
                    rhs.append(bk)
    J = sp.csr_matrix((vals, (rows, cols)), shape=(len(rhs), 4 * int(free.sum())))
    x_ls = spla.lsqr(J, np.array(rhs), atol=1e-14, btol=1e-14, iter_lim=50000)[0]
    s = oracle_solver(oracle_lib, P, nIterations=2, lIterations=600)
    s.solve(P.params)
    got = P.params[0][free].reshape(-1)
    assert rel_err(got, x_ls) < 1e-8
    assert np.array_equal(P.params[0][~free], X0[~free])            # excluded pixels never move


def test_image_warping_minimum_matches_scipy_least_squares(oracle_lib):
    """Nonlinear known answer: the image_warping energy written directly from the .t as a numpy residual function and
    minimised by scipy.optimize.least_squares must reach the same minimum as the oracle's Gauss-Newton (double)."""
    from scipy.optimize import least_squares
    P = wl.image_warping(7, 6, double=True, random_state=9, mask_fraction=0.1, perturb=0.3)
    O0, a0, U, C, M = [np.asarray(P.params[i], dtype=np.float64) for i in range(5)]
    wf, wr = float(P.params[5]), float(P.params[6])
    H, W = M.shape
    free = M == 0
    nfree = int(free.sum())

    def unpack(x):
        O, a = O0.copy(), a0.copy()
        O[free] = x[:2 * nfree].reshape(nfree, 2); a[free] = x[2 * nfree:]
        return O, a

    def residuals(x):
        O, a = unpack(x)
        out = []
        for y in range(H):
            for xx in range(W):
                if not free[y, xx]:
                    continue                                   # Exclude: rows centred on masked pixels are not in the cost
                c, s = np.cos(a[y, xx]), np.sin(a[y, xx])
                for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                    nx, ny = xx + dx, y + dy
                    if 0 <= nx < W and 0 <= ny < H and free[ny, nx]:
                        du = U[y, xx] - U[ny, nx]
                        rot = np.array([c * du[0] - s * du[1], s * du[0] + c * du[1]])
                        out.extend(wr * ((O[y, xx] - O[ny, nx]) - rot))
                if C[y, xx, 0] >= 0 and C[y, xx, 1] >= 0:
                    out.extend(wf * (O[y, xx] - C[y, xx]))
        return np.array(out)

    x0 = np.concatenate([O0[free].reshape(-1), a0[free]])
    sol = least_squares(residuals, x0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=20000)
    cost_ref = 0.5 * np.sum(sol.fun ** 2)
    s = oracle_solver(oracle_lib, P, nIterations=60, lIterations=200)
    assert abs(s.eval_cost(P.params) - 0.5 * np.sum(residuals(x0) ** 2)) <= 1e-12 * cost_ref + 1e-12   # same cost function
    s.solve(P.params)
    assert abs(s.cost() - cost_ref) <= 1e-7 * max(cost_ref, 1e-9)


def test_arap_cost_matches_numpy_restatement(oracle_lib):
    """arap_mesh_deformation.t evaluated with plain numpy (Rotate3D as in lib.t:77-91) must give the oracle's cost."""
    P = wl.arap_mesh_deformation(6, 5, double=True, seed=1, perturb=0.05)
    wf, wr, O, A, U, C, nE, v0, v1 = P.params
    cost = 0.0
    for v in range(P.dims[0]):
        if C[v, 0] >= -999999.9:
            cost += 0.5 * np.sum((float(wf) * (O[v] - C[v])) ** 2)
    for e in range(int(nE)):
        al, be, ga = A[v0[e]]
        ca, cb, cg, sa, sb, sg = np.cos(al), np.cos(be), np.cos(ga), np.sin(al), np.sin(be), np.sin(ga)
        R = np.array([[cg * cb, -sg * ca + cg * sb * sa, sg * sa + cg * sb * ca],
                      [sg * cb, cg * ca + sg * sb * sa, -cg * sa + sg * sb * ca],
                      [-sb, cb * sa, cb * ca]])
        r = float(wr) * ((O[v0[e]] - O[v1[e]]) - R @ (U[v0[e]] - U[v1[e]]))
        cost += 0.5 * np.sum(r ** 2)
    s = oracle_solver(oracle_lib, P)
    assert abs(s.eval_cost(P.params) - cost) <= 1e-12 * cost


def test_sfs_cost_matches_numpy_restatement(oracle_lib):
    """shape_from_shading.t evaluated with plain numpy -- B_I from workloads._sfs_render (the .t's equations 8/10 in array form),
    the three energy terms and the valid / interior gating written out -- must give the oracle's cost."""
    P = wl.shape_from_shading(18, 15, double=True, seed=3, holes=True, noise=2e-3)
    p = P.params
    w_p, w_s, w_g = (np.sqrt(float(p[i])) for i in range(3))
    fx, fy, ux, uy = (float(p[i]) for i in range(3, 7))
    L = [float(p[7 + i]) for i in range(9)]
    X, D, Im, mR, mC = p[16], p[17], p[18], p[19].astype(float), p[20].astype(float)
    H, W = X.shape
    B = wl._sfs_render(X, fx, fy, ux, uy, L)
    I = 0.5 * Im + 0.25 * (np.roll(Im, 1, axis=1) + np.roll(Im, 1, axis=0))
    dv = D > 0
    interior = np.zeros((H, W), bool); interior[1:-1, 1:-1] = True
    ok = interior & dv & np.roll(dv, 1, axis=1) & np.roll(dv, 1, axis=0)
    BI = np.where(ok, B - I, 0.0)
    cost = 0.0
    jj, ii = np.meshgrid(np.arange(H, dtype=float), np.arange(W, dtype=float), indexing="ij")
    Pts = np.stack([(ii - ux) / fx * X, (jj - uy) / fy * X, X], -1)
    for y in range(H):
        for x in range(W):
            if not dv[y, x]:
                continue                                                      # Exclude(D_i <= 0)
            cost += 0.5 * (w_p * (X[y, x] - D[y, x])) ** 2
            if interior[y, x]:
                cost += 0.5 * (w_g * (BI[y, x] - BI[y, x + 1]) * mR[y, x]) ** 2
                cost += 0.5 * (w_g * (BI[y, x] - BI[y + 1, x]) * mC[y, x]) ** 2
                nb = [(0, -1), (0, 1), (-1, 0), (1, 0)]
                valid = all(dv[y + dy, x + dx] for dx, dy in nb) and all(abs(X[y, x] - X[y + dy, x + dx]) < 0.01 for dx, dy in nb)
                if valid:
                    lap = 4.0 * Pts[y, x] - sum(Pts[y + dy, x + dx] for dx, dy in nb)
                    cost += 0.5 * np.sum((w_s * lap) ** 2)
    s = oracle_solver(oracle_lib, P)
    assert abs(s.eval_cost(P.params) - cost) <= 1e-10 * cost


def _bilinear(im, x, y):
    """Image:sample (o.t:578-589) restated with numpy index arithmetic: floor / ceil corners, zero outside the image."""
    H, W = im.shape
    x0, x1, y0, y1 = np.floor(x).astype(int), np.ceil(x).astype(int), np.floor(y).astype(int), np.ceil(y).astype(int)

    def get(xi, yi):
        ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
        return np.where(ok, im[np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)], 0.0)
    xn, yn = x - x0, y - y0
    return (1 - yn) * ((1 - xn) * get(x0, y0) + xn * get(x1, y0)) + yn * ((1 - xn) * get(x0, y1) + xn * get(x1, y1))


def test_optical_flow_cost_and_gradient_match_numpy_restatement(oracle_lib):
    """optical_flow.t:14-19 written out with whole-image numpy operations (independent of the per-residual oracle code)."""
    P = wl.optical_flow(13, 9, double=True, seed=5, init_flow=1.3)
    wf, wr, X, I, Ih, Dx, Dy = [np.asarray(a, dtype=np.float64) for a in P.params]
    H, W = I.shape
    xs, ys = np.meshgrid(np.arange(W, dtype=float), np.arange(H, dtype=float))
    px, py = xs + X[..., 0], ys + X[..., 1]
    fit = wf * (I - _bilinear(Ih, px, py))
    cost = 0.5 * (fit ** 2).sum()
    g = np.zeros_like(X)
    g[..., 0] = fit * (-wf * _bilinear(Dx, px, py)); g[..., 1] = fit * (-wf * _bilinear(Dy, px, py))
    for ax, sl_c, sl_n in ((1, np.s_[:, :-1], np.s_[:, 1:]), (0, np.s_[:-1], np.s_[1:])):
        d = wr * (X[sl_c] - X[sl_n])                 # each lattice edge carries two residuals (centred at either end), equal up to sign
        cost += 2 * 0.5 * (d ** 2).sum()
        g[sl_c] += 2 * wr * d; g[sl_n] -= 2 * wr * d
    s = oracle_solver(oracle_lib, P)
    assert abs(s.eval_cost(P.params) - cost) <= 1e-12 * cost
    f, _ = s.eval_jtf(P.params)
    assert rel_err(f, g.reshape(-1)) < 1e-12


def test_intrinsic_cost_matches_numpy_restatement(oracle_lib):
    """intrinsic_image_decomposition.t with p = 0.8: L_p weight sqrt((|dr| + 1e-7)^(p-2)) on the albedo differences (lib.t:106-114)."""
    P = wl.intrinsic_image_decomposition(11, 8, double=True, seed=6)
    wf, wa, ws, p, r, i, sh = [np.asarray(a, dtype=np.float64) for a in P.params]
    cost = 0.5 * ((wf * (r + sh[..., None] - i)) ** 2).sum()
    for sl_c, sl_n in ((np.s_[:, :-1], np.s_[:, 1:]), (np.s_[:-1], np.s_[1:])):
        dr = r[sl_c] - r[sl_n]
        wgt = np.sqrt((np.sqrt((dr ** 2).sum(-1)) + 1e-7) ** (p - 2))
        cost += 2 * 0.5 * ((wa * wgt[..., None] * dr) ** 2).sum()
        cost += 2 * 0.5 * ((ws * (sh[sl_c] - sh[sl_n])) ** 2).sum()
    s = oracle_solver(oracle_lib, P)
    assert abs(s.eval_cost(P.params) - cost) <= 1e-12 * cost


def test_volumetric_cost_matches_numpy_restatement(oracle_lib):
    """volumetric_mesh_deformation.t with Rotate3D (lib.t:77-91) as explicit rotation matrices built by numpy."""
    P = wl.volumetric_mesh_deformation(4, 5, 3, double=True, seed=8, perturb=0.1)
    O, A, U, C, wf, wr = [np.asarray(a, dtype=np.float64) for a in P.params]
    al, be, ga = A[..., 0], A[..., 1], A[..., 2]
    ca, cb, cg, sa, sb, sg = np.cos(al), np.cos(be), np.cos(ga), np.sin(al), np.sin(be), np.sin(ga)
    R = np.stack([np.stack([cg * cb, -sg * ca + cg * sb * sa, sg * sa + cg * sb * ca], -1),
                  np.stack([sg * cb, cg * ca + sg * sb * sa, -cg * sa + sg * sb * ca], -1),
                  np.stack([-sb, cb * sa, cb * ca], -1)], -2)                              # [..., row, col]
    valid = C[..., 0] >= -999999.9
    cost = 0.5 * ((wf * (O - np.where(np.isfinite(C), C, 0.0))) ** 2 * valid[..., None]).sum()
    for ax in range(3):
        for sgn in (1, -1):
            c = [slice(None)] * 3; n = [slice(None)] * 3
            c[ax], n[ax] = (slice(None, -1), slice(1, None)) if sgn == 1 else (slice(1, None), slice(None, -1))
            c, n = tuple(c), tuple(n)
            e = (O[c] - O[n]) - np.einsum("...ij,...j->...i", R[c], U[c] - U[n])
            cost += 0.5 * ((wr * e) ** 2).sum()
    s = oracle_solver(oracle_lib, P)
    assert abs(s.eval_cost(P.params) - cost) <= 1e-12 * cost


def _rot3(angles):
    """Rotate3D's matrix (lib.t:77-91) for an [n, 3] array of Euler angles -> [n, 3, 3]."""
    al, be, ga = angles[:, 0], angles[:, 1], angles[:, 2]
    ca, cb, cg, sa, sb, sg = np.cos(al), np.cos(be), np.cos(ga), np.sin(al), np.sin(be), np.sin(ga)
    return np.stack([np.stack([cg * cb, -sg * ca + cg * sb * sa, sg * sa + cg * sb * ca], -1),
                     np.stack([sg * cb, cg * ca + sg * sb * sa, -cg * sa + sg * sb * ca], -1),
                     np.stack([-sb, cb * sa, cb * ca], -1)], -2)


def test_cotangent_cost_matches_numpy_restatement(oracle_lib):
    """cotangent_mesh_smoothing.t written with whole-array numpy operations over the hyperedge list."""
    P = wl.cotangent_mesh_smoothing(6, 5, double=True, seed=4)
    wf, wr, X, A, nE, v0, v1, v2, v3 = [np.asarray(a) for a in P.params]
    X = X.astype(np.float64)

    def unit(d):
        return d / np.sqrt((d * d).sum(-1, keepdims=True))

    def cot(a, b):
        ab = (a * b).sum(-1)
        disc = (a * a).sum(-1) * (b * b).sum(-1) - ab * ab
        disc = np.where(disc > 0, disc, 0.0001)
        return ab / np.sqrt(disc)
    w = 0.5 * (cot(unit(X[v0] - X[v2]), unit(X[v1] - X[v2])) + cot(unit(X[v0] - X[v3]), unit(X[v1] - X[v3])))
    w = np.sqrt(np.where(w > 0, w, 0.0001))
    cost = 0.5 * ((float(wf) * (X - A)) ** 2).sum() + 0.5 * ((float(wr) * w[:, None] * (X[v1] - X[v0])) ** 2).sum()
    s = oracle_solver(oracle_lib, P)
    assert abs(s.eval_cost(P.params) - cost) <= 1e-12 * cost


def test_embedded_cost_matches_numpy_restatement(oracle_lib):
    P = wl.embedded_mesh_deformation(6, 4, double=True, seed=5, perturb=0.1)
    wf, wr, wrot, O, R, U, C, nE, v0, v1 = [np.asarray(a) for a in P.params]
    M = R.reshape(-1, 3, 3).astype(np.float64)                                   # row-major frames
    valid = C[:, 0] >= -999999.9
    cost = 0.5 * ((float(wf) * (O - np.where(np.isfinite(C), C, 0.0))) ** 2 * valid[:, None]).sum()
    G = np.einsum("nki,nkj->nij", M, M)                                          # columns' Gram matrix
    ortho = np.stack([G[:, 0, 1], G[:, 0, 2], G[:, 1, 2], G[:, 0, 0] - 1, G[:, 1, 1] - 1, G[:, 2, 2] - 1], -1)
    cost += 0.5 * ((float(wrot) * ortho) ** 2).sum()
    e = (O[v1] - O[v0]) - np.einsum("eij,ej->ei", M[v0], U[v1] - U[v0])
    cost += 0.5 * ((float(wr) * e) ** 2).sum()
    s = oracle_solver(oracle_lib, P)
    assert abs(s.eval_cost(P.params) - cost) <= 1e-12 * cost


def test_robust_alignment_cost_matches_numpy_restatement(oracle_lib):
    """robust_nonrigid_alignment.t: the scalar point-to-plane term and the confidence penalty are each selected by the three
    component-wise tests of greatereq(Constraints, -999999.9) (ad.t:327-349), i.e. counted once per valid component."""
    P = wl.robust_nonrigid_alignment(6, 5, double=True, seed=6, perturb=0.1)
    wf, wr, O, A, rw, U, C, Nn, nE, v0, v1 = [np.asarray(a) for a in P.params]
    valid = C >= -999999.9                                                        # [n, 3]
    Cz = np.where(np.isfinite(C), C, 0.0)
    fit = rw * (Nn * (O - Cz)).sum(-1)
    cost = 0.5 * ((float(wf) * fit[:, None]) ** 2 * valid).sum() + 0.5 * ((0.1 * (1 - rw * rw)[:, None]) ** 2 * valid).sum()
    e = (O[v0] - O[v1]) - np.einsum("eij,ej->ei", _rot3(A.astype(np.float64))[v0], U[v0] - U[v1])
    cost += 0.5 * ((float(wr) * e) ** 2).sum()
    s = oracle_solver(oracle_lib, P)
    assert abs(s.eval_cost(P.params) - cost) <= 1e-12 * cost
