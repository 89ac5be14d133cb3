"""CPU tests of the drop-in boundary (-m "not gpu"): the C-ABI library loads, exports every symbol the
headers declare, reads .t files, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(Opt(?:Amd)?_[A-Za-z]+)\s*\(", text)))


def test_headers_declare_the_reference_api():
    # reference API/release/include/Opt.h:35-71: exactly these ten entry points
    assert _declared_symbols("Opt.h") == sorted([
        "Opt_NewState", "Opt_ProblemDefine", "Opt_ProblemDelete", "Opt_ProblemPlan", "Opt_PlanFree",
        "Opt_SetSolverParameter", "Opt_ProblemSolve", "Opt_ProblemInit", "Opt_ProblemStep", "Opt_ProblemCurrentCost"])


def test_library_exports_every_declared_symbol(opt_lib):
    L = opt_lib.lib()
    for header in ("Opt.h", "OptAmd.h"):
        for sym in _declared_symbols(header):
            assert hasattr(L, sym), f"{sym} declared in include/{header} but not exported by libOpt.so"
    assert b"gfx950" in L.OptAmd_Version()


def test_initialization_parameter_struct_layout(opt_lib):
    # reference Opt.h:10-31: four ints, in this order
    P = opt_lib.Opt_InitializationParameters
    assert [f[0] for f in P._fields_] == ["doublePrecision", "verbosityLevel", "collectPerKernelTimingInfo", "threadsPerBlock"]
    assert ctypes.sizeof(P) == 16


def test_shipped_energy_files_match_registry(opt_lib):
    names = opt_lib.registered_energies()
    assert "image_warping" in names
    for n in names:
        ok, msg = opt_lib.check_problem_file(opt_lib.energy_file(n))
        assert ok, msg


def test_t_reader_rejects_mismatches(opt_lib, tmp_path):
    src = open(opt_lib.energy_file("image_warping")).read()
    # the file stem selects the kernel set; a KNOWN body saved under another name resolves by its content hash (reference o.t:840-853 loads
    # whatever path it is given); an unknown name with an unknown body has no kernel set
    p = tmp_path / "my_warp.t"; p.write_text(src)
    ok, msg = opt_lib.check_problem_file(str(p)); assert ok and "image_warping" in msg, msg
    p = tmp_path / "my_new_energy.t"; p.write_text(src.replace("eq(Mask(dx,dy), 0) * inShape", "inShape"))
    ok, msg = opt_lib.check_problem_file(str(p)); assert not ok and "no hand-written kernel set" in msg
    # binding index moved
    d = tmp_path / "a"; d.mkdir(); p = d / "image_warping.t"
    p.write_text(src.replace('Array("Mask", opt_float, {W,H}, 4)', 'Array("Mask", opt_float, {W,H}, 7)'))
    ok, msg = opt_lib.check_problem_file(str(p)); assert not ok and "Mask" in msg
    # type changed
    d = tmp_path / "b"; d.mkdir(); p = d / "image_warping.t"
    p.write_text(src.replace('Unknown("Angle", opt_float,', 'Unknown("Angle", opt_float2,'))
    ok, msg = opt_lib.check_problem_file(str(p)); assert not ok and "Angle" in msg
    # preconditioner flag changed
    d = tmp_path / "c"; d.mkdir(); p = d / "image_warping.t"
    p.write_text(src.replace("UsePreconditioner(true)", "UsePreconditioner(false)"))
    ok, msg = opt_lib.check_problem_file(str(p)); assert not ok and "UsePreconditioner" in msg
    # commented-out declarations are ignored, reformatting is fine
    d = tmp_path / "d"; d.mkdir(); p = d / "image_warping.t"
    p.write_text("-- Array(\"Bogus\", opt_float, {W,H}, 9)\n--[[ Param(\"x\", float, 11) ]]\n" + src.replace(", ", " ,  "))
    ok, msg = opt_lib.check_problem_file(str(p)); assert ok, msg
    # an edited energy BODY (declarations intact): the reference would compile the edit, hand-written kernels cannot honour it -> refused
    d = tmp_path / "e"; d.mkdir(); p = d / "image_warping.t"
    p.write_text(src.replace("Energy(w_fitSqrt * Select(hasTarget, Offset(0,0) - Constraints(0,0), 0.0))", ""))
    ok, msg = opt_lib.check_problem_file(str(p)); assert not ok and "energy body" in msg
    d = tmp_path / "f"; d.mkdir(); p = d / "image_warping.t"
    p.write_text(src.replace("eq(Mask(dx,dy), 0) * inShape", "inShape"))
    ok, msg = opt_lib.check_problem_file(str(p)); assert not ok and "energy body" in msg
    # missing file
    ok, msg = opt_lib.check_problem_file(str(tmp_path / "nope.t")); assert not ok and "cannot open" in msg


def test_energy_body_hash_table_is_current(opt_lib):
    """Every shipped .t hashes to an entry of opt_amd/csrc/energy_hashes.inc (tools/energy_hashes.py regenerates it after an edit)."""
    import ctypes
    L = opt_lib.lib()
    L.OptAmd_ProblemFileHash.restype = ctypes.c_ulong; L.OptAmd_ProblemFileHash.argtypes = [ctypes.c_char_p]
    table = open(os.path.join(ROOT, "opt_amd", "csrc", "energy_hashes.inc")).read()
    for n in opt_lib.registered_energies():
        h = L.OptAmd_ProblemFileHash(opt_lib.energy_file(n).encode())
        assert h != 0 and ('{"%s", 0x%016xul' % (n, h)) in table, n


def test_no_cpu_fallback_without_device(opt_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    # the product path must fail loudly, not fall back (Opt_NewState -> NULL, Solver raises)
    with pytest.raises(RuntimeError):
        opt_lib.Solver(opt_lib.energy_file("image_warping"), "gaussNewtonGPU", (8, 8))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under opt_amd/ may import, load or link it."""
    bad = re.compile(r"import\s+oracle|from\s+oracle|libopt_oracle|oracle/|OptOracle_")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "opt_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), os.path.join(dirpath, f)


def test_reference_energy_files_are_accepted(opt_lib):
    """Drop-in check, only where the reference checkout is mounted (this container; never on the GPU box): the .t files the
    reference's own examples pass to Opt_ProblemDefine must validate against the registered binding layouts unchanged."""
    import glob
    import os
    root = "/root/reference"
    if not os.path.isdir(root):
        pytest.skip("reference checkout not present")
    seen = 0
    for n in opt_lib.registered_energies():
        hits = glob.glob(os.path.join(root, "examples", "*", n + ".t")) + glob.glob(os.path.join(root, "tests", "*", n + ".t"))
        for h in hits:
            ok, msg = opt_lib.check_problem_file(h)
            assert ok, f"{h}: {msg}"
            seen += 1
    assert seen >= 8


def _prototypes(path):
    """{name: (return type, [parameter types])} of the Opt_* declarations in a header, comments and parameter names removed."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for ret, name, args in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(Opt_[A-Za-z]+)\s*\(([^)]*)\)\s*;", text):
        def norm(t):
            return re.sub(r"\s*\*\s*", "*", " ".join(t.split()))
        params = []
        for a in args.split(","):
            a = norm(a)
            m = re.match(r"^(.*?[\*\s])([A-Za-z_]\w*)$", a)          # drop the parameter name
            params.append(norm(m.group(1)) if m else a)
        out[name] = (norm(ret), params)
    return out


def test_prototypes_equal_the_reference_header_and_a_reference_caller_links(opt_lib, tmp_path):
    """Source and link compatibility with the reference's own header: same return and parameter types for all ten entry points, and a
    C caller compiled against /root/reference/API/release/include/Opt.h links against libOpt.so unchanged."""
    ref = "/root/reference/API/release/include/Opt.h"
    if not os.path.exists(ref):
        pytest.skip("reference checkout not present")
    mine = _prototypes(os.path.join(ROOT, "include", "Opt.h"))
    theirs = _prototypes(ref)
    assert sorted(mine) == sorted(theirs) and len(theirs) == 10
    for name in theirs:
        assert mine[name] == theirs[name], (name, mine[name], theirs[name])
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "caller.c"
    src.write_text('#include "Opt.h"\n'
                   "int main(int argc, char** argv) {\n"
                   "  if (argc < 100) return 0;            /* never runs: the test is that it compiles and links */\n"
                   "  struct Opt_InitializationParameters ip = {0, 0, 0, 0};\n"
                   "  Opt_State* s = Opt_NewState(ip);\n"
                   '  Opt_Problem* p = Opt_ProblemDefine(s, argv[1], "gaussNewtonGPU");\n'
                   "  unsigned int dims[2] = {4, 4}; void* params[1] = {0}; int n = 1;\n"
                   "  Opt_Plan* pl = Opt_ProblemPlan(s, p, dims);\n"
                   '  Opt_SetSolverParameter(s, pl, "nIterations", &n);\n'
                   "  Opt_ProblemInit(s, pl, params); while (Opt_ProblemStep(s, pl, params)) {}\n"
                   "  Opt_ProblemSolve(s, pl, params);\n"
                   "  double c = Opt_ProblemCurrentCost(s, pl);\n"
                   "  Opt_PlanFree(s, pl); Opt_ProblemDelete(s, p);\n"
                   "  return c > 0;\n}\n")
    libdir = os.path.dirname(opt_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-I", os.path.dirname(ref), str(src), "-L", libdir, "-lOpt", "-Wl,-rpath," + libdir, "-o", str(tmp_path / "caller")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_bench_helpers_without_a_gpu():
    """bench.py's bookkeeping that needs no device: the kernel-source hash is stable and names real files, the frozen oracle values of the metric's solve load,
    the reference-spread yardstick comes out of the committed golden files, and a traffic file is only accepted for the kernel sources it was measured with."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    for f in bench.KERNEL_SOURCES:
        assert os.path.exists(os.path.join(ROOT, f)), f
    sha = bench.kernel_src_sha16()
    assert len(sha) == 16 and sha == bench.kernel_src_sha16()
    fl, db = bench.golden_solve8(2048)
    assert fl is not None and len(fl) == 9 and fl[0] == 35515200.0 and fl[-1] < 1e-3 * fl[0]
    rs = bench.reference_spread()
    y = rs.yardstick("horizon_2048_float_400", "float")
    assert 1e-4 < y < 1e-2            # 2.3e-3: diameter of ten legal runs (reference-order sums under five seeds, exact-order sums, plain / fma build) after 400 float iterations
    assert rs.n_reference_order_runs("horizon_2048_float_400") >= 5
    assert bench.measured_traffic("0" * 16) == (None, None)
    costs, env = bench.golden_cost(4096, 400)
    assert costs is not None and len(costs) == 3 and env is not None and env[0] == 0.0
