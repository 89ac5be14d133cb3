"""Build libOpt.so (the HIP product library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as well as on the MI355X box.
The built .so stays under opt_amd/lib/ (git-ignored, but it travels with gpurun snapshots).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libOpt.so")
OBJDIR = os.path.join(HERE, "build")
# -fno-slp-vectorize: the SLP vectoriser pairs unrelated fp32 operations into v_pk_* instructions and pays for it with
# v_mov packing (and, in the row-marching kernels, with copies of not-yet-arrived loads); measured 1-3 % slower.
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-misleading-indentation"]
# every device object is compiled with the kernel-resource-usage remarks on (free: an analysis pass the back end runs anyway); they are kept next to the object as
# <source>.remarks and read by tests/test_kernel_resources.py (VGPRs, scratch, occupancy, LDS of every kernel in the library -- the "no scratch in an offered variant" rule)
REMARKS = ["-Rpass-analysis=kernel-resource-usage"]


def kernel_resources():
    """{demangled kernel name: {"vgprs", "agprs", "scratch", "occupancy", "lds", "sgprs", "file"}} of every kernel of libOpt.so, from the remarks of the last build."""
    import re
    out = {}
    for path in sorted(glob.glob(os.path.join(OBJDIR, "*.remarks"))):
        txt = open(path, errors="replace").read()
        rows = [m.groups() for m in re.finditer(r"Function Name: (\S+).*?SGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", txt, re.S)]
        if not rows:
            continue
        names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
        for r, n in zip(rows, names):
            n = re.sub(r"\(.*", "", n.replace("optamd::(anonymous namespace)::", "").replace("void ", ""))
            out[n] = {"sgprs": int(r[1]), "vgprs": int(r[2]), "agprs": int(r[3]), "scratch": int(r[4]), "occupancy": int(r[5]), "lds": int(r[6]), "file": os.path.basename(path)[:-len(".o.remarks")]}
    return out


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _deps_mtime():
    hs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return max(os.path.getmtime(h) for h in hs)


def build_variant(name, defines):
    """Development aid: build opt_amd/lib/libOpt_<name>.so with extra -D defines (entries starting with '-' are passed
    to hipcc verbatim); select it at run time with OPT_AMD_LIB=<path>."""
    objdir = os.path.join(HERE, "build_" + name)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + [d if d.startswith("-") else "-D" + d for d in defines] + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    lib = os.path.join(LIBDIR, f"libOpt_{name}.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_m = _deps_mtime()
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        objs.append(obj)
        is_hip = src.endswith(".hip")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m) or (is_hip and not os.path.exists(obj + ".remarks")):
            cmd = [HIPCC] + FLAGS + (REMARKS + ["-x", "hip"] if is_hip else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stderr=open(obj + ".remarks", "w") if is_hip else None), obj + ".remarks" if is_hip else None))
    failed = []
    for src, p, rem in procs:
        if p.wait() != 0:
            failed.append(src)
            if rem:      # the compiler's messages went to the remarks file: show what is not a remark
                sys.stderr.write("".join(l for l in open(rem, errors="replace") if "remark:" not in l))
                os.remove(rem)
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if procs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_comm(force=force, verbose=verbose)
    build_examples(force=force, verbose=verbose)
    return LIB


EXAMPLES_DIR = os.path.join(HERE, "..", "examples")
EXAMPLES = ["minimal_laplacian", "minimal_graph_only", "create_delete_cycle", "image_warping_example", "poisson_example", "arap_example", "sfs_example"]


def build_examples(force=False, verbose=False):
    """C++ callers of the C ABI (examples/*.cpp -> examples/bin/*), linked against libOpt.so like a reference user would."""
    bindir = os.path.join(EXAMPLES_DIR, "bin")
    os.makedirs(bindir, exist_ok=True)
    outs, procs = [], []
    hdr = max(os.path.getmtime(os.path.join(EXAMPLES_DIR, "common.h")), os.path.getmtime(os.path.join(HERE, "..", "include", "Opt.h")))
    for name in EXAMPLES:
        src, out = os.path.join(EXAMPLES_DIR, name + ".cpp"), os.path.join(bindir, name)
        outs.append(out)
        if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), hdr, os.path.getmtime(LIB)):
            cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(HERE, "..", "include"), src, "-o", out,
                   "-L" + LIBDIR, "-lOpt", "-Wl,-rpath," + LIBDIR]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    failed = [s for s, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    return outs


COMM_LIB = os.path.join(LIBDIR, "libOptComm.so")


def build_comm(force=False, verbose=False):
    """libOptComm.so: the peer-mailbox (HIP kernels over IPC-mapped windows), RCCL and in-process implementations of
    OptAmd_SlabComm (multi-GPU slab tiling)."""
    src = os.path.join(CSRC, "comm", "opt_comm.cpp")
    peer = os.path.join(CSRC, "comm", "peer_comm.hip")
    if force or not os.path.exists(COMM_LIB) or os.path.getmtime(COMM_LIB) < max(os.path.getmtime(src), os.path.getmtime(peer), _deps_mtime()):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", COMM_LIB, src, "-x", "hip", peer, "-L/opt/rocm/lib", "-lrccl", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return COMM_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
