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
    # unknown energy (file stem selects the kernel set)
    p = tmp_path / "my_new_energy.t"; p.write_text(src)
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
    # missing file
    ok, msg = opt_lib.check_problem_file(str(tmp_path / "nope.t")); assert not ok and "cannot open" in msg


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
