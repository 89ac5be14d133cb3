"""CPU test (no GPU): register / scratch budget of every kernel in libOpt.so, read from the compiler's own resource remarks.

opt_amd/build.py compiles every device object with -Rpass-analysis=kernel-resource-usage and keeps the remarks next to the object.  The rule: a kernel that the
runtime can OFFER runs without scratch -- the persistent ("on-chip") kernels are all latency, a scratch reload inside their loop sits on the critical path; the marching
kernels are sized against the register file.  Variants that would spill are not instantiated at all (Op::spills / Op::kMaxBlock); two image_warping on-chip variants
are listed with the bytes they still hold and why, so that a regression (or an improvement) shows up here instead of in a profile.
"""
import re

import pytest

from opt_amd import build


@pytest.fixture(scope="module")
def resources(opt_lib):
    build.build()      # (re)compiles whatever has no remarks file yet
    r = build.kernel_resources()
    assert len(r) > 300, len(r)
    return r


# kernels that still hold scratch, with the ceiling asserted here: {regex on the demangled name: (max bytes per lane, why)}
KNOWN = {
    r"^iw_onchipPcg<float, 16, true, true, false>$": (40, "the 4096x512 slab variant: p, r, cos / sin of 16 rows = 132 persistent VGPRs, A p fills 96 of the 160 KB of LDS and the rest is "
                                                          "taken to within 6 KB, delta is streamed through L2 -- ~13 scratch operations per iteration of ~2000 VALU instructions"),
    r"^iw_onchipPcg<float, 8, false, false, true>$": (76, "Levenberg-Marquardt, 8 rows: p, r, delta, A p of 8 rows in registers; cos / sin moved to LDS in round 6 (124 -> 68 B; 76 with the early-out report for verbose callers), b already there; "
                                                          "the LDS is full"),
}


def test_no_offered_kernel_uses_scratch(resources):
    bad = []
    for name, r in sorted(resources.items()):
        if r["scratch"] == 0:
            continue
        for pat, (cap, _why) in KNOWN.items():
            if re.search(pat, name):
                if r["scratch"] > cap:
                    bad.append((name, r["scratch"], f"known, but above its ceiling of {cap}"))
                break
        else:
            bad.append((name, r["scratch"], "not in the list of known exceptions"))
    assert not bad, bad


def test_known_exceptions_still_exist(resources):
    """(an exception that no longer spills must leave the list: the list is documentation)"""
    for pat in KNOWN:
        hit = [n for n in resources if re.search(pat, n)]
        assert hit and all(resources[n]["scratch"] > 0 for n in hit), (pat, [(n, resources[n]["scratch"]) for n in hit])


def test_persistent_kernels_leave_room_for_their_workgroup(resources):
    """A persistent kernel's workgroup must fit a CU by itself: waves per SIMD needed by the workgroup <= occupancy the register budget allows."""
    n = 0
    for name, r in resources.items():
        m = re.match(r"^(iw_onchipPcg|sfs_onchipPcg|march_onchipPcg)<(.*)>$", name)
        if not m:
            continue
        n += 1
        args = [a.strip() for a in m.group(2).split(",")]
        if m.group(1) == "iw_onchipPcg":
            waves = 8
        elif m.group(1) == "sfs_onchipPcg":
            waves = int(args[3])
        else:
            waves = int(args[-2])
        assert r["occupancy"] * 4 >= waves, (name, r)
        assert r["vgprs"] + r["agprs"] <= 512 // (waves // 4), (name, r)      # 512 registers per SIMD lane, waves / 4 waves of the workgroup per SIMD
    assert n > 100, n


def test_the_benchmarked_kernel_has_headroom(resources):
    """iw_pcgIter2<float, unit lattice, table preconditioner, Gauss-Newton> (bench.py's PCGIteration): 768-thread workgroups need 3 waves per SIMD -> at most 168 VGPRs, no scratch."""
    hits = {n: r for n, r in resources.items() if re.match(r"^iw_pcgIter2<float, true, 3, (true|false), false, [012]>$", n)}
    assert hits
    for n, r in hits.items():
        assert r["scratch"] == 0 and r["vgprs"] <= 168 and r["occupancy"] >= 3, (n, r)
