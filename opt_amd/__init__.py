"""opt_amd -- MI355X-native Gauss-Newton / Levenberg-Marquardt solver backend behind niessner/Opt's C API.

    api        ctypes mirror of include/Opt.h + include/OptAmd.h (libOpt.so, HIP only -- no CPU fallback)
    build      hipcc build of libOpt.so / libOptComm.so / the C++ example callers (gfx950)
    slab       row-slab tiling of image problems over the GPUs of a node (RCCL or in-process threads)
    workloads  synthetic instances of the registered energies (host numpy)
    io         readers / writers of the reference examples' input formats
"""
__version__ = "0.1.0"
