"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU oracle (oracle/libopt_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (opt_amd/) never does.  The call shape mirrors Opt.h (reference API/release/include/Opt.h:35-71)
with HOST arrays in `params`.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libopt_oracle.so")
_lib = None
# OPT_ORACLE_VARIANT=fma: the control build with fused multiply-adds (oracle/Makefile); only tests/golden/make_horizon_costs.py uses it
if os.environ.get("OPT_ORACLE_VARIANT") == "fma":
    _LIB_PATH = os.path.join(_HERE, "libopt_oracle_fma.so")


def build(force=False):
    """Compile the oracle with g++ (no GPU, no reference sources involved)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("oracle_capi.cpp", "solver.hpp", "energies.hpp", "sfs.hpp", "dual.hpp")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", os.path.basename(_LIB_PATH)])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, cp, ci, cd, cl = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_double, ctypes.c_long
        L.OptOracle_Create.restype = vp
        L.OptOracle_Create.argtypes = [cp, cp, ci, ctypes.POINTER(ctypes.c_uint)]
        L.OptOracle_Free.argtypes = [vp]
        L.OptOracle_SetSolverParameter.restype = ci
        L.OptOracle_SetSolverParameter.argtypes = [vp, cp, vp]
        for f in ("OptOracle_Init", "OptOracle_Solve"):
            getattr(L, f).argtypes = [vp, ctypes.POINTER(vp)]
        L.OptOracle_Step.restype = ci
        L.OptOracle_Step.argtypes = [vp, ctypes.POINTER(vp)]
        L.OptOracle_CurrentCost.restype = cd
        L.OptOracle_CurrentCost.argtypes = [vp]
        L.OptOracle_SetThreads.argtypes = [vp, ci]
        L.OptOracle_SetReduction.argtypes = [vp, ci, ctypes.c_uint]
        L.OptOracle_SetTrigVariant.argtypes = [ctypes.c_uint]
        L.OptOracle_NumUnknownScalars.restype = cl
        L.OptOracle_NumUnknownScalars.argtypes = [vp]
        L.OptOracle_GetVector.restype = ci
        L.OptOracle_GetVector.argtypes = [vp, cp, vp]
        L.OptOracle_EvalJTF.argtypes = [vp, ctypes.POINTER(vp), vp, vp]
        L.OptOracle_ApplyJTJ.argtypes = [vp, ctypes.POINTER(vp), vp, vp]
        L.OptOracle_EvalCost.restype = cd
        L.OptOracle_EvalCost.argtypes = [vp, ctypes.POINTER(vp)]
        L.OptOracle_TraceRows.restype = cl
        L.OptOracle_TraceRows.argtypes = [vp]
        L.OptOracle_GetTrace.argtypes = [vp, vp]
        L.OptOracle_CostHistoryLen.restype = cl
        L.OptOracle_CostHistoryLen.argtypes = [vp]
        L.OptOracle_GetCostHistory.argtypes = [vp, vp]
        L.OptOracle_TrustRegionRadius.restype = cd
        L.OptOracle_TrustRegionRadius.argtypes = [vp]
        L.OptOracle_PoissonPatchSolve.argtypes = [ci, ci, ci, vp, vp, vp, ci, ci, ci, ci, vp]
        _lib = L
    return _lib


def set_trig_variant(seed):
    """Process-wide: 0 = float sin / cos of the host libm (default); n > 0 = a seeded stand-in for another implementation within 1 ulp (oracle/dual.hpp:
    the reference calls libdevice's, the HIP product ocml's -- legal elementwise variants of the same algorithm)."""
    lib().OptOracle_SetTrigVariant(int(seed))


_INT_PARAMS = {"nIterations", "lIterations", "residual_reset_period", "nIter"}


def _param_array(params):
    """params: list of numpy arrays (host); returns (void*[] , keepalive)."""
    arr = (ctypes.c_void_p * len(params))()
    keep = []
    for i, p in enumerate(params):
        a = np.ascontiguousarray(p) if not (isinstance(p, np.ndarray) and p.flags["C_CONTIGUOUS"]) else p
        keep.append(a)
        arr[i] = a.ctypes.data
    return arr, keep


class OracleSolver:
    def __init__(self, energy, kind="gaussNewtonGPU", double=False, dims=(1, 1)):
        d = (ctypes.c_uint * len(dims))(*[int(x) for x in dims])
        self._h = lib().OptOracle_Create(energy.encode(), kind.encode(), int(bool(double)), d)
        if not self._h:
            raise ValueError(f"oracle: unknown energy {energy!r} or solver kind {kind!r}")
        self.dtype = np.float64 if double else np.float32
        self.n = lib().OptOracle_NumUnknownScalars(self._h)

    def close(self):
        if self._h:
            lib().OptOracle_Free(self._h)
            self._h = None

    __del__ = close

    def set(self, name, value):
        v = np.array(value, dtype=np.int32 if name in _INT_PARAMS else np.float32)
        ok = lib().OptOracle_SetSolverParameter(self._h, name.encode(), v.ctypes.data)
        if not ok:
            raise KeyError(name)

    def set_threads(self, n):
        """OpenMP threads for the timed CPU baseline (row bands); parity tests keep the default of 1."""
        lib().OptOracle_SetThreads(self._h, int(n))

    def set_reduction(self, mode, seed=0):
        """0: long-double sums rounded once (default).  1: the reference's own reduction arithmetic -- opt_float terms per element, the 32-lane
        shfl.down tree per warp, one opt_float atomicAdd per warp in an order drawn from `seed` (oracle/solver.hpp header)."""
        lib().OptOracle_SetReduction(self._h, int(mode), int(seed))

    def init(self, params):
        a, self._keep = _param_array(params)
        lib().OptOracle_Init(self._h, a)

    def step(self, params):
        a, self._keep = _param_array(params)
        return lib().OptOracle_Step(self._h, a)

    def solve(self, params):
        a, self._keep = _param_array(params)
        lib().OptOracle_Solve(self._h, a)

    def cost(self):
        return lib().OptOracle_CurrentCost(self._h)

    def vector(self, name):
        out = np.zeros(self.n, dtype=self.dtype)
        if not lib().OptOracle_GetVector(self._h, name.encode(), out.ctypes.data):
            raise KeyError(name)
        return out

    def eval_jtf(self, params):
        a, keep = _param_array(params)
        f = np.zeros(self.n, dtype=self.dtype)
        d = np.zeros(self.n, dtype=self.dtype)
        lib().OptOracle_EvalJTF(self._h, a, f.ctypes.data, d.ctypes.data)
        return f, d

    def apply_jtj(self, params, v):
        a, keep = _param_array(params)
        v = np.ascontiguousarray(v, dtype=self.dtype)
        out = np.zeros(self.n, dtype=self.dtype)
        lib().OptOracle_ApplyJTJ(self._h, a, v.ctypes.data, out.ctypes.data)
        return out

    def eval_cost(self, params):
        a, keep = _param_array(params)
        return lib().OptOracle_EvalCost(self._h, a)

    def trace(self):
        n = lib().OptOracle_TraceRows(self._h)
        out = np.zeros((n, 6), dtype=np.float64)
        if n:
            lib().OptOracle_GetTrace(self._h, out.ctypes.data)
        return out

    def cost_history(self):
        n = lib().OptOracle_CostHistoryLen(self._h)
        out = np.zeros(n, dtype=np.float64)
        if n:
            lib().OptOracle_GetCostHistory(self._h, out.ctypes.data)
        return out

    def trust_region_radius(self):
        return lib().OptOracle_TrustRegionRadius(self._h)


def poisson_patch_solve(X, T, M, n_iterations, l_iterations, patch_iterations=16, patch_size=32):
    """Block-local patch solver of oracle/patch.hpp.  X (H, W, 4), T (H, W, 4), M (H, W), all float32 or all float64.
    Returns (X after the solve, costs[n_iterations + 1])."""
    dt = X.dtype
    assert dt in (np.float32, np.float64) and T.dtype == dt and M.dtype == dt
    H, W = M.shape
    Xo = np.ascontiguousarray(X).copy()
    Tc, Mc = np.ascontiguousarray(T), np.ascontiguousarray(M)
    costs = np.zeros(n_iterations + 1, dtype=np.float64)
    lib().OptOracle_PoissonPatchSolve(int(dt == np.float64), W, H, Xo.ctypes.data, Tc.ctypes.data, Mc.ctypes.data, int(n_iterations), int(l_iterations),
                                      int(patch_iterations), int(patch_size), costs.ctypes.data)
    return Xo, costs
