"""ctypes binding of libOpt.so -- the Python mirror of Opt.h (reference API/release/include/Opt.h:35-71).

Function names, argument order and meaning are the C API's: Opt_NewState, Opt_ProblemDefine,
Opt_ProblemPlan, Opt_SetSolverParameter, Opt_ProblemInit / Step / Solve, Opt_ProblemCurrentCost,
Opt_PlanFree, Opt_ProblemDelete.  `Solver` bundles them the way the reference's example harness does
(examples/shared/OptSolver.h:40-97).  Device memory is the caller's: torch-ROCm tensors are passed by
data_ptr(); scalar Params and graph edge counts are host numpy scalars, as in the reference.

There is no CPU fallback: loading fails loudly if libOpt.so is missing, and Opt_NewState returns NULL
without a HIP device.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OPT_AMD_LIB") or os.path.join(_HERE, "lib", "libOpt.so")   # OPT_AMD_LIB: development A/B builds
_lib = None

OPT_SYMBOLS = ["Opt_NewState", "Opt_ProblemDefine", "Opt_ProblemDelete", "Opt_ProblemPlan", "Opt_PlanFree",
               "Opt_SetSolverParameter", "Opt_ProblemSolve", "Opt_ProblemInit", "Opt_ProblemStep", "Opt_ProblemCurrentCost"]
OPTAMD_SYMBOLS = ["OptAmd_Version", "OptAmd_EnergyCount", "OptAmd_EnergyName", "OptAmd_PlanNumUnknownScalars", "OptAmd_PlanVector",
                  "OptAmd_EvalJTF", "OptAmd_ApplyJTJ", "OptAmd_EvalCost", "OptAmd_PlanEnableTrace", "OptAmd_PlanTraceRows",
                  "OptAmd_PlanGetTrace", "OptAmd_PlanTrustRegionRadius", "OptAmd_PlanKernelTiming", "OptAmd_PlanSetTiming", "OptAmd_PlanKernelCount",
                  "OptAmd_PlanKernelName", "OptAmd_PlanOnChipStatus", "OptAmd_PlanDescribe", "OptAmd_PlanSetSlab", "OptAmd_PlanSetSlabExt", "OptAmd_CheckProblemFile", "OptAmd_ProblemFileHash",
                  "OptAmd_MeasureCopyBandwidth", "OptAmd_DebugOccupy"]


class Opt_InitializationParameters(ctypes.Structure):
    _fields_ = [("doublePrecision", ctypes.c_int), ("verbosityLevel", ctypes.c_int),
                ("collectPerKernelTimingInfo", ctypes.c_int), ("threadsPerBlock", ctypes.c_int)]


HALO_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                           ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_long), ctypes.c_void_p)
ALLREDUCE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)
ALLREDUCE_PARTIALS_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)


class OptAmd_SlabComm(ctypes.Structure):
    _fields_ = [("ctx", ctypes.c_void_p), ("rank", ctypes.c_int), ("world", ctypes.c_int),
                ("haloExchange", HALO_FN), ("allReduceSum", ALLREDUCE_FN)]


class OptAmd_SlabCommExt(ctypes.Structure):      # include/OptAmd.h: optional accelerations, versioned by its leading size field
    _fields_ = [("size", ctypes.c_ulong), ("allReducePartials", ALLREDUCE_PARTIALS_FN), ("allReducePost", ctypes.c_void_p), ("allReducePlan", ctypes.c_void_p),
                ("onChipPlan", ctypes.c_void_p)]


def lib():
    """Load libOpt.so (built in-tree by opt_amd.build).  Raises if it is missing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -m opt_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    vp, cp, ci, cd, cl = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_double, ctypes.c_long
    pvp = ctypes.POINTER(vp)
    L.Opt_NewState.restype = vp; L.Opt_NewState.argtypes = [Opt_InitializationParameters]
    L.Opt_ProblemDefine.restype = vp; L.Opt_ProblemDefine.argtypes = [vp, cp, cp]
    L.Opt_ProblemDelete.restype = None; L.Opt_ProblemDelete.argtypes = [vp, vp]
    L.Opt_ProblemPlan.restype = vp; L.Opt_ProblemPlan.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_uint)]
    L.Opt_PlanFree.restype = None; L.Opt_PlanFree.argtypes = [vp, vp]
    L.Opt_SetSolverParameter.restype = None; L.Opt_SetSolverParameter.argtypes = [vp, vp, cp, vp]
    L.Opt_ProblemSolve.restype = None; L.Opt_ProblemSolve.argtypes = [vp, vp, pvp]
    L.Opt_ProblemInit.restype = None; L.Opt_ProblemInit.argtypes = [vp, vp, pvp]
    L.Opt_ProblemStep.restype = ci; L.Opt_ProblemStep.argtypes = [vp, vp, pvp]
    L.Opt_ProblemCurrentCost.restype = cd; L.Opt_ProblemCurrentCost.argtypes = [vp, vp]
    L.OptAmd_Version.restype = cp
    L.OptAmd_EnergyCount.restype = ci
    L.OptAmd_EnergyName.restype = cp; L.OptAmd_EnergyName.argtypes = [ci]
    L.OptAmd_PlanNumUnknownScalars.restype = cl; L.OptAmd_PlanNumUnknownScalars.argtypes = [vp]
    L.OptAmd_PlanVector.restype = vp; L.OptAmd_PlanVector.argtypes = [vp, cp]
    L.OptAmd_EvalJTF.restype = None; L.OptAmd_EvalJTF.argtypes = [vp, vp, pvp, vp, vp]
    L.OptAmd_ApplyJTJ.restype = cd; L.OptAmd_ApplyJTJ.argtypes = [vp, vp, pvp, vp, vp]
    L.OptAmd_EvalCost.restype = cd; L.OptAmd_EvalCost.argtypes = [vp, vp, pvp]
    L.OptAmd_PlanEnableTrace.restype = None; L.OptAmd_PlanEnableTrace.argtypes = [vp, ci]
    L.OptAmd_PlanTraceRows.restype = cl; L.OptAmd_PlanTraceRows.argtypes = [vp]
    L.OptAmd_PlanGetTrace.restype = None; L.OptAmd_PlanGetTrace.argtypes = [vp, vp]
    L.OptAmd_PlanTrustRegionRadius.restype = cd; L.OptAmd_PlanTrustRegionRadius.argtypes = [vp]
    L.OptAmd_PlanOnChipStatus.restype = ci; L.OptAmd_PlanOnChipStatus.argtypes = [vp]
    L.OptAmd_PlanDescribe.restype = ci; L.OptAmd_PlanDescribe.argtypes = [vp, ctypes.c_char_p, ci]
    L.OptAmd_PlanSetTiming.restype = None; L.OptAmd_PlanSetTiming.argtypes = [vp, ci]
    L.OptAmd_PlanKernelTiming.restype = ci; L.OptAmd_PlanKernelTiming.argtypes = [vp, cp, ctypes.POINTER(cl), ctypes.POINTER(cd)]
    L.OptAmd_PlanKernelCount.restype = ci; L.OptAmd_PlanKernelCount.argtypes = [vp]
    L.OptAmd_PlanKernelName.restype = cp; L.OptAmd_PlanKernelName.argtypes = [vp, ci]
    L.OptAmd_PlanSetSlab.restype = ci; L.OptAmd_PlanSetSlab.argtypes = [vp, cl, cl, cl, ctypes.POINTER(OptAmd_SlabComm)]
    L.OptAmd_PlanSetSlabExt.restype = ci; L.OptAmd_PlanSetSlabExt.argtypes = [vp, ctypes.POINTER(OptAmd_SlabCommExt)]
    L.OptAmd_MeasureCopyBandwidth.restype = cd; L.OptAmd_MeasureCopyBandwidth.argtypes = [cl, ci, ci]
    L.OptAmd_DebugOccupy.restype = ci; L.OptAmd_DebugOccupy.argtypes = [ci, cd, vp]
    L.OptAmd_CheckProblemFile.restype = ci; L.OptAmd_CheckProblemFile.argtypes = [cp, cp, ci]
    _lib = L
    return L


def check_problem_file(path):
    """(ok, message): host-only validation of a .t file against the kernel registry."""
    buf = ctypes.create_string_buffer(512)
    ok = lib().OptAmd_CheckProblemFile(path.encode(), buf, 512)
    return bool(ok), buf.value.decode()


_INT_PARAMS = {"nIterations", "lIterations", "residual_reset_period", "nIter", "patchIterations", "patchSize", "amd_reference_order", "amd_onchip"}


def energy_file(stem):
    """Path of the .t file shipped with this package for energy `stem`."""
    return os.path.join(_HERE, "energies", stem + ".t")


def registered_energies():
    L = lib()
    return [L.OptAmd_EnergyName(i).decode() for i in range(L.OptAmd_EnergyCount())]


class ParamPack:
    """problemparams (void**) built from torch device tensors and host numpy scalars."""

    def __init__(self, params):
        self.keep = list(params)
        self.array = (ctypes.c_void_p * len(params))()
        for i, p in enumerate(params):
            if hasattr(p, "data_ptr"):          # torch tensor (device array / index array)
                self.array[i] = p.data_ptr()
            elif isinstance(p, np.ndarray):     # host scalar (Param / graph edge count)
                self.array[i] = p.ctypes.data
            else:
                raise TypeError(f"problem parameter {i}: expected a torch tensor or a numpy array, got {type(p)}")


class Solver:
    """One Opt state + problem + plan, like the reference harness's OptSolver (examples/shared/OptSolver.h:40-97)."""

    def __init__(self, filename, kind, dims, double=False, verbosity=0, timing=False):
        L = lib()
        ip = Opt_InitializationParameters(int(bool(double)), int(verbosity), int(bool(timing)), 0)
        self.state = L.Opt_NewState(ip)
        if not self.state:
            raise RuntimeError("Opt_NewState failed (no HIP device?)")
        self.problem = L.Opt_ProblemDefine(self.state, filename.encode(), kind.encode())
        d = (ctypes.c_uint * len(dims))(*[int(x) for x in dims])
        self.plan = L.Opt_ProblemPlan(self.state, self.problem, d)
        if not self.plan:
            L.Opt_ProblemDelete(self.state, self.problem)
            self.problem = None
            raise RuntimeError(f"Opt_ProblemPlan returned NULL for {filename!r} ({kind})")
        self.double = bool(double)
        self._pack = None

    # -- Opt.h ------------------------------------------------------------------------------------------
    def set_parameter(self, name, value):
        v = np.array(value, dtype=np.int32 if name in _INT_PARAMS else np.float32)
        lib().Opt_SetSolverParameter(self.state, self.plan, name.encode(), v.ctypes.data)

    def _params(self, params):
        self._pack = params if isinstance(params, ParamPack) else ParamPack(params)
        return self._pack.array

    def init(self, params):
        lib().Opt_ProblemInit(self.state, self.plan, self._params(params))

    def step(self, params):
        return lib().Opt_ProblemStep(self.state, self.plan, self._params(params))

    def solve(self, params):
        lib().Opt_ProblemSolve(self.state, self.plan, self._params(params))

    def cost(self):
        return lib().Opt_ProblemCurrentCost(self.state, self.plan)

    def close(self):
        if getattr(self, "plan", None):
            lib().Opt_PlanFree(self.state, self.plan)      # teardown order of OptSolver.h:59-68
            self.plan = None
        if getattr(self, "problem", None):
            lib().Opt_ProblemDelete(self.state, self.problem)
            self.problem = None

    __del__ = close

    # -- OptAmd.h probes --------------------------------------------------------------------------------
    @property
    def n(self):
        return lib().OptAmd_PlanNumUnknownScalars(self.plan)

    def vector(self, name):
        """Copy of a solver vector as a torch tensor on the current device."""
        import torch
        ptr = lib().OptAmd_PlanVector(self.plan, name.encode())
        if not ptr:
            raise KeyError(name)
        dt = torch.float64 if self._is_double() else torch.float32
        out = torch.empty(self.n, dtype=dt, device="cuda")
        es = 8 if dt == torch.float64 else 4
        torch.cuda.synchronize()
        _hip_memcpy_d2d(out.data_ptr(), ptr, self.n * es)
        return out

    def _is_double(self):
        return self.double and not getattr(self, "float_only", False)

    def eval_jtf(self, params):
        import torch
        dt = torch.float64 if self._is_double() else torch.float32
        f = torch.zeros(self.n, dtype=dt, device="cuda")
        d = torch.zeros(self.n, dtype=dt, device="cuda")
        torch.cuda.synchronize()
        lib().OptAmd_EvalJTF(self.state, self.plan, self._params(params), f.data_ptr(), d.data_ptr())
        return f, d

    def apply_jtj(self, params, v):
        import torch
        out = torch.zeros_like(v)
        torch.cuda.synchronize()
        dot = lib().OptAmd_ApplyJTJ(self.state, self.plan, self._params(params), v.data_ptr(), out.data_ptr())
        return out, dot

    def eval_cost(self, params):
        import torch
        torch.cuda.synchronize()
        return lib().OptAmd_EvalCost(self.state, self.plan, self._params(params))

    def enable_trace(self, on=True):
        lib().OptAmd_PlanEnableTrace(self.plan, int(on))

    def trace(self):
        n = lib().OptAmd_PlanTraceRows(self.plan)
        out = np.zeros((n, 6))
        if n:
            lib().OptAmd_PlanGetTrace(self.plan, out.ctypes.data)
        return out

    def trust_region_radius(self):
        return lib().OptAmd_PlanTrustRegionRadius(self.plan)

    def describe(self):
        """What the plan would do at its next step (OptAmd_PlanDescribe), as a dict of the kernel set's key=value pairs."""
        buf = ctypes.create_string_buffer(2048)
        lib().OptAmd_PlanDescribe(self.plan, buf, 2048)
        return dict(kv.strip().split("=", 1) for kv in buf.value.decode().split(";") if "=" in kv)

    def on_chip_status(self):
        """0: launch-per-iteration kernels; 1: the last step's linear solve ran as one on-chip launch; 2: an on-chip wait timed out, the plan fell back for good."""
        return lib().OptAmd_PlanOnChipStatus(self.plan)

    def set_timing(self, on):
        """Per-kernel hipEvent timing on / off from the next launch on (totals restart).  on = 2: one event pair per run of consecutive launches of one name."""
        lib().OptAmd_PlanSetTiming(self.plan, int(on))

    def kernel_timings(self):
        """{kernel name: (count, total_ms)} since the last init (needs timing=True)."""
        L = lib()
        out = {}
        for i in range(L.OptAmd_PlanKernelCount(self.plan)):
            name = L.OptAmd_PlanKernelName(self.plan, i)
            c, t = ctypes.c_long(0), ctypes.c_double(0)
            L.OptAmd_PlanKernelTiming(self.plan, name, ctypes.byref(c), ctypes.byref(t))
            out[name.decode()] = (c.value, t.value)
        return out


_hip = None


def _hip_memcpy_d2d(dst, src, nbytes):
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _hip.hipMemcpy.restype = ctypes.c_int
    err = _hip.hipMemcpy(dst, src, nbytes, 3)   # hipMemcpyDeviceToDevice
    if err != 0:
        raise RuntimeError(f"hipMemcpy failed: {err}")


def to_device(problem):
    """Upload a workloads.Problem: arrays -> torch device tensors, host scalars stay numpy."""
    import torch
    out = []
    for p in problem.params:
        a = np.asarray(p)
        if a.ndim == 0:
            out.append(np.array(a))                 # host scalar (Param / edge count)
        else:
            out.append(torch.from_numpy(np.ascontiguousarray(a)).cuda())
    return out
