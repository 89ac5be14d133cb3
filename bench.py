#!/usr/bin/env python
"""bench.py -- PCG iterations/s of the Gauss-Newton solve of image_warping 4096^2 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--size 4096]

A "step" is one Opt_ProblemStep: one Gauss-Newton iteration = evalJTF + `lIterations` (400, the reference's
examples/image_warping/src/main.cpp:113-114) matrix-free PCG iterations + update + cost.
value = K * lIterations / wall time of the K timed steps (max over ranks), inputs resident in HBM.

N > 1: one rank per GPU.  Under torch.distributed.run (how the driver launches it) the ranks are already there; a plain
`python bench.py --gpus N` re-executes itself under torch.distributed.run with N ranks.  The image is split into row slabs with 8
ghost rows; per PCG iteration the ranks all-reduce four doubles (peer-mapped mailbox over xGMI, opt_amd/csrc/comm; RCCL with
OPT_AMD_COMM=rccl) and every 7th iteration exchange 8 edge rows of r and p with each slab neighbour -- total work fixed => "strong" scaling.
`--size 8192` is BASELINE config 5.

The same JSON line carries
  roofline     : the dominant kernel (`PCGIteration`: ONE launch per PCG iteration doing the work of the reference's PCGStep1 + PCGStep2 +
                 PCGStep3) timed with hipEvents on the solver's stream.  `achieved` = the bytes the kernel has to move (its own byte model,
                 53 B/pixel: the last two search directions in, the new one out, angle, flags, delta every second launch -- DESIGN.md section 3.1) / average launch time,
                 `frac` = achieved / 8 TB/s: a physical fraction.  `traffic` = HBM bytes per launch measured with rocprofv3 PMC passes of
                 THIS kernel source (profiles/*_traffic.json carries the source hash; a stale file is ignored), `hbm_frac` = traffic / time / peak.
                 `algorithmic_equiv` keeps SURVEY.md 8(d)'s scale: the reference algorithm's 180 B/pixel (three kernels) over the same time.
                 N > 1: after the timed steps the same job runs two more steps with per-kernel hipEvents switched on (OptAmd_PlanSetTiming); rank 0 reports
                 its own slab's kernel (model bytes of its rows / its launch time) and `per_iteration_ms`: the difference is the communicator (all-reduce
                 every iteration, halo exchange every 7th) plus launch gaps -- on a real multi-GPU box that is the xGMI cost per iteration.
  contract_loop: the reference-ordered loop as a product mode (Opt_SetSolverParameter "amd_reference_order" = 1: PCGStep1; PCGStep2; PCGStep3 per iteration, r / z / A p in
                 memory) on the same workload: PCG iterations/s, per-kernel times, `frac` against the reference formulation's 180 B/pixel and `frac_physical` against the 168.8 it
                 moves (PMC-measured), relative error per step against the frozen exact-order oracle builds -- what parity at the contract costs.
  smoke        : N > 1 only: a Gauss-Newton step of 12 PCG iterations through the real kernels on every rank BEFORE the timed region, verdict collective; a peer communicator
                 that fails it is replaced by RCCL on every rank and the line says so.
  cpu_baseline : the CPU oracle (a port, not the reference) timed on a bounded sample on the host cores (rank 0, N = 1 only).
  parity       : cost after the first step next to the frozen oracle value for this workload (tests/golden/bench_costs.json).
  box          : what the line ran on -- power cap, clocks and power sampled under load, the box's own measured copy bandwidth (boxes of the pool differ by ~10 %).
  --dry        : set-up only; prints what every rank WOULD do (on chip or streaming, communicator, ghost depth, bytes per exchange) and exits.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

# dmabuf IPC (hipIpcGetMemHandle across processes): must be in the environment before the HIP runtime initialises, i.e. before torch is imported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PIXEL = 48 + 96 + 36   # PCGStep1 + PCGStep2 + PCGStep3 of the reference formulation
# what iw_pcgIter2 has to move per pixel per launch, float, Gauss-Newton: p_{k-1} 12 + p_{k-2} 12 in (r is rebuilt from them),
# p_k 12 out, angle 4, flags 1 = 41; every second launch additionally delta 12 in / 12 out = 24 -> 12 on average; general UrShape: + U 8 + M_a 4
MODEL_BYTES_PER_PIXEL = {"lattice": 41 + 12, "general": 41 + 12 + 8}      # general UrShape: + U 8 (M_O from the flag byte; M_a rebuilt from the pairs the stencil evaluates, round 4)
HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CONTRACT_LOOP_TRAFFIC_B_PER_PX = 168.8      # reference-ordered loop, PMC-measured per PCG iteration at 4096^2 (profiles/r06_contract_loop_traffic.txt): k_step2 96.0 + iw_applyJTJ<fused> 72.8
KERNEL_SOURCES = ["opt_amd/csrc/energy_image_warping.hip", "opt_amd/csrc/iw_device.h", "opt_amd/csrc/iw_iter.h", "opt_amd/csrc/iw_step.h", "opt_amd/csrc/iw_onchip.h", "opt_amd/csrc/solver.hip", "opt_amd/csrc/common.h", "opt_amd/csrc/energy.h", "opt_amd/build.py"]


def kernel_src_sha16():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--liters", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline / general-UrShape legs (profiling runs)")
    ap.add_argument("--cpu-size", type=int, default=4096)
    ap.add_argument("--cpu-liters", type=int, default=24)      # ~12 s of host work on the 128-thread box (the contract asks for 10-30 s)
    ap.add_argument("--comm", default=os.environ.get("OPT_AMD_COMM", "peer"), choices=["peer", "rccl"])
    ap.add_argument("--cpu-smoke", action="store_true", help="launcher check without GPUs: ranks rendezvous over gloo and report the world size")
    ap.add_argument("--dry", action="store_true", help="set everything up (slabs, communicator, plan) and print, per rank, what the solve WOULD do -- on chip or one launch per "
                                                       "iteration, communicator and its fast paths, ghost depth, bytes per exchange (OptAmd_PlanDescribe) -- without running a step")
    ap.add_argument("--share-gpu", action="store_true", help="functional check of the N-rank path on a 1-GPU box: all ranks use device 0, set-up over gloo (timings meaningless)")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch(n):
    """`python bench.py --gpus N` without a launcher: run N ranks of this script under torch.distributed.run."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(size, liters):
    """Oracle (CPU restatement of the reference algorithm, OpenMP over row bands) on a bounded sample of the workload."""
    from oracle.binding import OracleSolver
    from opt_amd import workloads as wl
    cores = os.cpu_count() or 1
    phys = None
    try:      # physical cores = distinct (socket, core) pairs: SMT siblings buy the memory-bound row bands nothing, so one thread per physical core (at most 128)
        ids, cur = set(), {}
        for ln in open("/proc/cpuinfo"):
            if ":" in ln:
                k, v = [t.strip() for t in ln.split(":", 1)]
                cur[k] = v
            elif cur:
                ids.add((cur.get("physical id"), cur.get("core id"))); cur = {}
        if cur:
            ids.add((cur.get("physical id"), cur.get("core id")))
        phys = len(ids) if len(ids) > 1 else None
    except OSError:
        pass
    threads = max(1, min(phys or cores, 128))
    P = wl.image_warping(size, size)
    s = OracleSolver("image_warping", "gaussNewtonGPU", False, P.dims)
    s.set_threads(threads)
    s.set("nIterations", 1); s.set("lIterations", liters)
    s.init(P.params)
    t0 = time.perf_counter()
    s.step(P.params)
    dt = time.perf_counter() - t0
    rate = liters / dt * (size * size) / (4096.0 * 4096.0)     # scaled to 4096^2-equivalent PCG iterations/s
    return {"value": rate, "unit": "PCG iters/s", "cores": threads, "kind": "port",
            "sample": f"oracle (C++ port of solverGPUGaussNewton.t: generic dual-number residuals scattered per thread band, {threads} OpenMP threads), "
                      f"image_warping {size}x{size} float, 1 GN step x {liters} PCG iterations, {dt:.1f} s wall incl. the step's evalJTF/update/cost; "
                      f"host has {cores} logical / {phys or '?'} physical cores: one thread per physical core (SMT siblings do not help the memory-bound row bands), capped at 128.  "
                      "A stated baseline, not a tuned CPU solver: no speed-up claim is made from it"}


def golden_cost(size, liters):
    """(oracle float costs, float-rounding envelope per step) for this workload from tests/golden/bench_costs.json, or (None, None)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "bench_costs.json")) as f:
            G = json.load(f)
        fl = G[f"image_warping_{size}x{size}_float_gaussNewtonGPU_{liters}"]["costs"]
    except (OSError, KeyError):
        return None, None
    db = G.get(f"image_warping_{size}x{size}_double_gaussNewtonGPU_{liters}", {}).get("costs")
    env = [abs(a - b) / abs(b) for a, b in zip(fl, db)] if db else None
    return fl, env


def golden_solve8(size):
    """Frozen oracle costs of the metric's solve (8 GN steps x 400 PCG iterations from the initial guess): (float run, double run) or (None, None)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "horizon_costs.json")) as f:
            G = json.load(f)
    except OSError:
        return None, None
    fl, db = G.get(f"solve8_{size}_float"), G.get(f"solve8_{size}_double")
    return (fl["costs"] if fl else None), (db["costs"] if db else None)


def reference_spread():
    """tools/reference_spread.py: the one yardstick of long-horizon parity -- the diameter of the frozen legal runs of the reference's arithmetic (reference-order
    sums under several seeds, exact-order sums, plain / fma build of the oracle).  Reads tests/golden/ only."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import reference_spread as rs
    return rs


def measured_traffic(sha):
    """HBM bytes per launch of the iteration kernel from the newest profiles/*_traffic.json taken with this kernel source."""
    import glob
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        tj = json.load(open(cand))
        if tj.get("bench_kernel") == "PCGIteration" and tj.get("kernel_src_sha16") == sha:
            return tj["hbm_bytes_per_launch"], os.path.basename(cand)
    return None, None


def _pcg_rate(api, wl, torch, W, H, liters, onchip, steps=3):
    os.environ["OPT_AMD_ONCHIP"] = "1" if onchip else "0"
    try:
        P = wl.image_warping(W, H)
        g = api.Solver(api.energy_file(P.energy), "gaussNewtonGPU", P.dims)
        g.set_parameter("nIterations", steps + 1); g.set_parameter("lIterations", liters)
        dev = api.to_device(P)
        g.init(dev); g.step(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.step(dev)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        g.close()
        return dt
    finally:
        del os.environ["OPT_AMD_ONCHIP"]


def onchip_table(api, wl, torch, liters):
    """PCG iterations/s where the whole linear solve runs as one persistent launch with its state on chip -- the reference's own input sizes (512^2:
    examples/image_warping/src/main.cpp:98-134; the SFS fixture's 640x480), 1024^2, and 2 M pixels = 1/8 of the metric's 4096^2 -- next to the streaming
    loop (one launch per PCG iteration, OPT_AMD_ONCHIP=0) on the same box."""
    rows = []
    for W, H in [(512, 512), (640, 480), (1024, 1024), (2048, 1024), (4096, 512)]:
        t_on, t_st = _pcg_rate(api, wl, torch, W, H, liters, True), _pcg_rate(api, wl, torch, W, H, liters, False)
        rows.append({"image": f"{W}x{H}", "pixels": W * H, "onchip_us_per_iter": 1e6 * t_on / liters, "onchip_pcg_iters_per_s": liters / t_on,
                     "streaming_us_per_iter": 1e6 * t_st / liters, "streaming_pcg_iters_per_s": liters / t_st, "speedup": t_st / t_on})
    return {"what": "image_warping float, Gauss-Newton steps of %d PCG iterations, wall time per step / %d; on-chip = iw_onchipPcg (one launch per linear solve), "
                    "streaming = iw_pcgIter2 (one launch per PCG iteration)" % (liters, liters),
            "sizes": rows, "onchip_4096x512_us_per_iter": rows[-1]["onchip_us_per_iter"], "streaming_4096x512_us_per_iter": rows[-1]["streaming_us_per_iter"]}


def reference_example_flows():
    """The reference's example programs as a user runs them, through the C++ callers of the C ABI (examples/*.cpp): image_warping 512^2, 19 constraint passes x 8
    Gauss-Newton x 400 PCG (examples/image_warping/src/main.cpp:110-134; GN and LM on identical inputs), and shape_from_shading 640x480 double LM 60 x 10
    (examples/shape_from_shading/src/main.cpp:27-38) on a procedural surface with the fixture's parameters.  Solver time = sum of the per-step wall times the
    harness records (its results CSV)."""
    import re
    out = {}
    exe = os.path.join(ROOT, "examples", "bin", "image_warping_example")
    sfs = os.path.join(ROOT, "examples", "bin", "sfs_example")
    if not (os.path.exists(exe) and os.path.exists(sfs)):
        return None

    def run(cmd, env_extra):
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        return r.returncode, r.stdout

    for name, extra in (("onchip", {}), ("streaming", {"OPT_AMD_ONCHIP": "0"})):
        rc, txt = run([exe, "512", "19", "8", "400"], extra)
        m = re.search(r"total solver time: GN ([0-9.]+) ms, LM ([0-9.]+) ms", txt)
        c = re.search(r"Opt GN,Opt LM,CERES\s*\n([-+0-9.eE]+),([-+0-9.eE]+),", txt)
        out["image_warping_512_19x8x400_" + name] = {"rc": rc, "gn_ms": float(m.group(1)) if m else None, "lm_ms": float(m.group(2)) if m else None,
                                                      "final_cost_gn": float(c.group(1)) if c else None, "final_cost_lm": float(c.group(2)) if c else None}
    rc, txt = run([sfs, "-", "640", os.path.join(ROOT, "opt_amd", "energies", "shape_from_shading.t"), "480"], {})
    m = re.search(r"total solver time: ([0-9.]+) ms", txt)
    c = re.search(r"Opt GN,Opt LM,CERES\s*\n,([-+0-9.eE]+),", txt)
    out["shape_from_shading_640x480_60x10_double_lm"] = {"rc": rc, "lm_ms": float(m.group(1)) if m else None, "final_cost_lm": float(c.group(1)) if c else None}
    for f in ("results_float.csv", "results_double.csv"):
        try:
            os.remove(os.path.join(ROOT, f))
        except OSError:
            pass
    return out


def contract_loop_leg(api, wl, torch, W, H, liters):
    """The reference-ordered loop as a product mode (Opt_SetSolverParameter "amd_reference_order" = 1: PCGStep1; PCGStep2; PCGStep3 per iteration as
    solverGPUGaussNewton.t:1056-1092 runs them, r / z / A p in memory, the beta numerator summed directly) on the SAME workload as the headline: what parity at the
    contract costs.  Timed like the headline (wall clock over whole Opt_ProblemSteps), then one more step with one hipEvent pair per run of equally named launches.
    `frac` is a PHYSICAL fraction here: the loop moves the reference formulation's 180 B/pixel per iteration.  rel_err: cost after the first step and after
    the metric's 8 x 400 solve against the frozen exact-order oracle of the fused-multiply-add build (the legal run this loop restates) and the plain build."""
    P = wl.image_warping(W, H)
    dev = api.to_device(P)
    s = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", (W, H))
    s.set_parameter("amd_reference_order", 1)
    s.set_parameter("nIterations", 8); s.set_parameter("lIterations", liters)
    desc = s.describe()
    s.init(dev)
    costs = [s.cost()]
    s.step(dev); costs.append(s.cost())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        s.step(dev); costs.append(s.cost())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    s.set_timing(2)
    s.step(dev); costs.append(s.cost())
    torch.cuda.synchronize()
    kt = s.kernel_timings()
    s.set_timing(False)
    while s.step(dev):
        costs.append(s.cost())
    torch.cuda.synchronize()
    on_chip = s.on_chip_status()
    s.close()
    del dev
    loop_ms = sum(v[1] for k, v in kt.items() if k.startswith("PCGStep"))
    achieved = ALGO_BYTES_PER_PIXEL * W * H / (loop_ms / liters * 1e-3) / 1e9 if loop_ms else None
    out = {"what": f"image_warping {W}x{H} float, gaussNewtonGPU, {liters} PCG iterations per step, Opt_SetSolverParameter(amd_reference_order = 1)",
           "plan": desc, "pcg_iters_per_s": liters / dt, "ms_per_step": dt * 1e3, "on_chip_status": on_chip,
           "kernel_avg_ms": {k: v[1] / v[0] for k, v in kt.items()}, "kernel_ms_per_step": {k: v[1] for k, v in kt.items()},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved else None,
                        "bytes_per_pixel": ALGO_BYTES_PER_PIXEL, "us_per_iteration": 1e3 * loop_ms / liters if loop_ms else None,
                        "physical_bytes_per_pixel": CONTRACT_LOOP_TRAFFIC_B_PER_PX, "frac_physical": (achieved * CONTRACT_LOOP_TRAFFIC_B_PER_PX / ALGO_BYTES_PER_PIXEL / HBM_PEAK_GBS) if achieved else None,
                        "note": "SURVEY 8d's algorithmic bytes (PCGStep1 48 + PCGStep2 96 + PCGStep3 36 B/px) over the loop's own kernel time (PCGStep* launches of one step / "
                                "lIterations).  The loop keeps r, z, A p and p in memory and sums as the reference does; the one fusion left is PCGStep3 into the next PCGStep1 "
                                "(p is not written and re-read in between).  The bytes that physically move were measured with rocprofv3 FETCH_SIZE / WRITE_SIZE passes "
                                "(profiles/r06_contract_loop_traffic.txt, 4096^2): PCGStep2 96.0 B/px (= its model), PCGStep3+PCGStep1 72.8 B/px (model 53 + the cos / sin table 8 + "
                                "stencil halo re-reads) = 168.8 B/px: frac_physical"},
           "costs": costs}
    try:
        G = {"plain": json.load(open(os.path.join(ROOT, "tests", "golden", "horizon_costs.json"))), "fma": json.load(open(os.path.join(ROOT, "tests", "golden", "horizon_costs_fma.json")))}
        for name, g in G.items():
            e = g.get(f"solve8_{W}_float")
            if e and liters == 400 and len(costs) == len(e["costs"]):
                out[f"rel_err_vs_{name}_oracle"] = {"after_1_step_400_pcg": abs(costs[1] - e["costs"][1]) / abs(e["costs"][1]), "after_8_steps": abs(costs[-1] - e["costs"][-1]) / abs(e["costs"][-1]),
                                                    "per_step": [abs(a - b) / abs(b) for a, b in zip(costs, e["costs"])]}
        if "rel_err_vs_fma_oracle" in out:
            out["within_contract_1e-5_of_fma_oracle"] = max(out["rel_err_vs_fma_oracle"]["per_step"]) <= 1e-5
    except OSError:
        pass
    return out


def box_info(torch):
    """Which box is this?  Boxes of the pool differ by ~10 % on the same binary (sclk / power behaviour): the line says what it ran on.  Static facts from rocm-smi
    (power cap, mclk, performance level), and -- because an idle GPU reports its sleep clocks -- a one-second copy loop during which sclk / mclk / power are sampled,
    whose own GB/s is the box's measured-copy ceiling (MI355X_MICROARCH.md quotes 6.29 TB/s)."""
    import re
    import threading
    info = {}

    def smi(*opts):
        try:
            return subprocess.run(["rocm-smi", *opts], capture_output=True, text=True, timeout=20).stdout
        except Exception as e:      # noqa
            return f"rocm-smi failed: {e}"

    def grab(txt, pat):
        m = re.search(pat, txt)
        return float(m.group(1)) if m else None

    static = smi("--showmaxpower", "--showperflevel")
    info["power_cap_w"] = grab(static, r"Max Graphics Package Power \(W\): ([0-9.]+)")
    m = re.search(r"Performance Level: (\w+)", static)
    info["perf_level"] = m.group(1) if m else None
    n = 1 << 28      # 1 GiB of floats in, 1 GiB out
    a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    b.copy_(a); torch.cuda.synchronize()
    sample = {}
    th = threading.Thread(target=lambda: sample.update(txt=smi("--showclocks", "--showpower")))
    t0 = time.perf_counter()
    th.start()
    reps = 0
    while time.perf_counter() - t0 < 1.0:
        for _ in range(8):
            b.copy_(a)
        torch.cuda.synchronize()
        reps += 8
    dt = time.perf_counter() - t0
    th.join()
    info["copy_gbs"] = 2.0 * n * 4 * reps / dt / 1e9
    del a, b
    torch.cuda.empty_cache()
    # the ceiling as MI355X_MICROARCH.md measures it: a float4 grid-stride copy kernel (libOpt's own, OptAmd_MeasureCopyBandwidth), 2 GiB moved per launch, default and
    # nontemporal accesses -- `copy_gbs` above is torch's copy_ (round 5's probe), kept for comparison between rounds
    from opt_amd import api as _api
    info["copy_gbs_float4"] = _api.lib().OptAmd_MeasureCopyBandwidth(2 << 30, 0, 40)
    info["copy_gbs_float4_nontemporal"] = _api.lib().OptAmd_MeasureCopyBandwidth(2 << 30, 1, 40)
    a = torch.empty(1 << 25, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    # the same copy on a working set that stays in the 256 MiB Infinity Cache (64 MiB in + 64 MiB out): the ceiling a streaming kernel can reach on the cache-resident
    # working sets of configs 3 and 4 (SFS 1024^2 double: 132 MB; ARAP 500 k vertices: 239 MB) -- measured, not quoted
    m = 1 << 24
    for _ in range(4):
        b[:m].copy_(a[:m])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(200):
        b[:m].copy_(a[:m])
    torch.cuda.synchronize()
    info["copy_gbs_cache_resident_128MiB"] = 2.0 * m * 4 * 200 / (time.perf_counter() - t1) / 1e9
    txt = sample.get("txt", "")
    info["sclk_mhz_under_load"] = grab(txt, r"sclk clock level: \S+ \(([0-9.]+)Mhz\)")
    info["mclk_mhz_under_load"] = grab(txt, r"mclk clock level: \S+ \(([0-9.]+)Mhz\)")
    info["power_w_under_load"] = grab(txt, r"Current Socket Graphics Package Power \(W\): ([0-9.]+)")
    info["device"] = torch.cuda.get_device_name(0)
    info["kind"] = f"copy {info['copy_gbs'] / 1e3:.2f} TB/s (torch copy_), float4 kernel {info['copy_gbs_float4'] / 1e3:.2f} / nontemporal {info['copy_gbs_float4_nontemporal'] / 1e3:.2f} TB/s, sclk {info['sclk_mhz_under_load']} MHz under load, cap {info['power_cap_w']} W"
    del a, b
    return info


def flow_parity(flows):
    """The example flow's final costs (both HIP paths) next to the frozen oracle runs of the SAME flow (tests/golden/reference_flow_costs.json: exact-order sums and
    reference-order seeds, tests/golden/make_reference_flow.py): where do the HIP paths land relative to the runs of the reference's own arithmetic?"""
    try:
        G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_flow_costs.json")))
    except OSError:
        return None
    out = {}
    for kind, field in (("gaussNewtonGPU", "final_cost_gn"), ("LMGPU", "final_cost_lm")):
        e = G.get(f"image_warping_512_float_{kind}_19x8x400")
        if not e:
            continue
        runs = {seed: c[-1] for seed, c in e["costs_by_seed"].items()}
        vals = sorted(runs.values())
        med = vals[len(vals) // 2] if len(vals) % 2 else 0.5 * (vals[len(vals) // 2 - 1] + vals[len(vals) // 2])
        row = {"frozen_runs": runs, "frozen_median": med, "frozen_min": vals[0], "frozen_max": vals[-1], "frozen_diameter_rel": (vals[-1] - vals[0]) / med}
        for name in ("onchip", "streaming"):
            f = (flows or {}).get("image_warping_512_19x8x400_" + name)
            if f and f.get(field) is not None:
                c = f[field]
                row["hip_" + name] = {"final_cost": c, "rel_to_median": (c - med) / med, "inside_frozen_range": vals[0] <= c <= vals[-1],
                                      "within_contract_1e-5_of_some_run": any(abs(c - v) <= 1e-5 * abs(v) for v in vals)}
        out[kind] = row
    out["note"] = ("seed 0 = exact-order sums, seeds >= 1 = the reference's own sums (float atomics in a seeded order): every frozen run is a legal run of the reference; "
                   "their diameter is what the reference's 1e-5 contract can mean over 19 x 8 x 400 float iterations")
    return out


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(relaunch(args.gpus))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); they must agree")
    distributed = world > 1

    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    if args.cpu_smoke:      # launcher / rendezvous check on a box without GPUs (tests/test_slab_cpu.py)
        if distributed:
            dist.init_process_group("gloo")
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            print(json.dumps({"cpu_smoke": True, "n_gpus": world, "ranks_seen": seen}))
        return

    from opt_amd import api, build, workloads as wl
    if args.share_gpu:
        local_rank = 0
        # all ranks on one GPU: their iteration kernels poll each other's posted sums, so together they must fit the chip (opt_amd/csrc/energy_image_warping.hip splitRows)
        os.environ.setdefault("OPT_AMD_ITER_MAXWG", str(max(1, 224 // max(1, world))))
        os.environ.setdefault("OPT_AMD_PEER_POST", "1")      # (the communicator takes the posted all-reduce away when ranks share a GPU, unless told that the grids are capped)
    torch.cuda.set_device(local_rank)
    if distributed:
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0 and not os.path.exists(api.LIB_PATH):
        build.build()
    if distributed:
        dist.barrier()          # nobody loads libOpt.so before rank 0 has finished building it

    W = H = args.size
    total_steps = args.warmup + args.steps

    comm_ranks = 1
    preflight = None
    if distributed:
        from opt_amd import slab
        job = slab.SlabJob("image_warping", W, H, rank, world, comm=args.comm)     # every rank generates only its own slab (+ ghost rows)
        solver, dev, host0, unknown_slots = job.solver, job.params, job.local.params, job.local.unknown_slots
        comm_ranks = job.comm_ranks()
        args.comm = job.comm_kind              # "rccl" if the peer communicator was unavailable or failed its self-test on this machine
        preflight = [None] * world             # per rank: which devices it can reach, IPC window / open / self-test, which communicator was chosen and why
        dist.all_gather_object(preflight, job.preflight)
    else:
        P = wl.image_warping(W, H)
        dev = api.to_device(P)
        host0, unknown_slots = P.params, P.unknown_slots
        solver = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", (W, H))
    extra_steps = 0 if args.no_extras else 2      # the roofline leg runs on the SAME plan, after the timed steps (per-kernel hipEvents switched on by OptAmd_PlanSetTiming)
    solver.set_parameter("nIterations", total_steps + extra_steps)
    solver.set_parameter("lIterations", args.liters)

    if args.dry:      # what WOULD this run do?  One line, every rank's answer in it; no step is taken
        mine = {"rank": rank, "device": torch.cuda.current_device(), "plan": solver.describe(), "comm": args.comm if distributed else None,
                "slab": {"row0": job.layout.row0, "rows": job.layout.rows, "ghost": job.layout.ghost} if distributed else None}
        allr = [None] * world
        if distributed:
            dist.all_gather_object(allr, mine)
        else:
            allr = [mine]
        if rank == 0:
            print(json.dumps({"dry": True, "n_gpus": world, "workload": f"image_warping {W}x{H} float, gaussNewtonGPU, {args.liters} PCG iterations per GN step", "ranks": allr, "preflight": preflight}))
        if distributed:
            job.close()
            dist.destroy_process_group()
        else:
            solver.close()
        return

    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- N > 1: a smoke solve BEFORE the timed region.  The communicator's own self-test (one all-reduce, one halo exchange) has passed by
    # now; this runs the real thing -- one Gauss-Newton step of 12 PCG iterations through the solver's kernels: the posted all-reduce polled by the next launch's prologue,
    # the deep-ghost exchange, or (slabs that fit the chip) the on-chip solve with its edge boxes and rank hop -- and makes the verdict collective.  A rank that reports a
    # communicator error, a non-finite cost, or costs that differ between ranks sends EVERY rank to a fresh RCCL communicator; the line says so (`smoke`).  Every wait
    # involved is bounded (kernel-side 2 s, communicator OPT_AMD_PEER_TIMEOUT), so a peer path that does not work across real devices costs seconds, not the run.
    smoke = None
    if distributed:
        def smoke_step(j):
            j.solver.set_parameter("nIterations", 1); j.solver.set_parameter("lIterations", 12)
            t1 = time.perf_counter()
            try:
                j.solver.init(j.params); j.solver.step(j.params)
                torch.cuda.synchronize()
                c, err, exc = j.solver.cost(), j.comm_error(), None
            except Exception as e:      # noqa
                c, err, exc = float("nan"), -1, repr(e)
            mine = {"rank": rank, "cost": c, "comm_error": err, "exception": exc, "on_chip_status": j.solver.on_chip_status() if exc is None else None, "ms": 1e3 * (time.perf_counter() - t1)}
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            ok = all(r["comm_error"] == 0 and r["exception"] is None and r["cost"] == r["cost"] and abs(r["cost"] - allr[0]["cost"]) <= 1e-6 * abs(allr[0]["cost"]) for r in allr)
            return ok, allr
        ok, ranks = smoke_step(job)
        smoke = {"comm": job.comm_kind, "ok": ok, "ranks": ranks}
        if not ok and job.comm_kind == "peer" and not args.share_gpu:      # (ranks that share one GPU cannot form an RCCL communicator: the failure is only reported)
            if rank == 0:
                print("bench.py: the peer communicator failed the smoke solve; every rank switches to RCCL", file=sys.stderr, flush=True)
            try:
                job.close()
            except Exception:      # noqa
                pass
            job = slab.SlabJob("image_warping", W, H, rank, world, comm="rccl")
            solver, dev, host0, unknown_slots = job.solver, job.params, job.local.params, job.local.unknown_slots
            comm_ranks = job.comm_ranks(); args.comm = job.comm_kind
            ok2, ranks2 = smoke_step(job)
            smoke["fallback"] = {"comm": "rccl", "ok": ok2, "ranks": ranks2}
        for slot in unknown_slots:      # the smoke step moved the unknowns: back to the initial guess
            dev[slot].copy_(torch.from_numpy(host0[slot]))
        solver.set_parameter("nIterations", total_steps + extra_steps)
        solver.set_parameter("lIterations", args.liters)

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def comm_error():
        """Non-zero if ANY rank's communicator is in its error state (a wait for a peer timed out, an oversize exchange, a HIP / RCCL error): the sums of that
        run are garbage.  Collective, so that every rank takes the same exit."""
        if not distributed:
            return 0
        t = torch.tensor([float(job.comm_error())], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())

    def fail(code, where):
        """One JSON error line on rank 0 and a non-zero exit code on every rank (the driver reads stdout; a library exit(1) used to leave nothing)."""
        if rank == 0:
            print(json.dumps({"error": "communicator failure", "where": where, "code": code, "codes": "peer: 1 all-reduce timed out, 2 halo acknowledgement, 3 halo rows, "
                              "4 posted all-reduce polled by the iteration kernel, 5 oversize exchange, 6 too many values, 7 HIP error; rccl: ncclResult_t",
                              "n_gpus": world, "comm": args.comm, "metric": f"PCG iters/s, GN solve of image_warping {W}^2", "value": None}), flush=True)
        if distributed:
            try:
                dist.destroy_process_group()
            except Exception:      # noqa
                pass
        os._exit(4)

    # ---- the timed region: K Opt_ProblemSteps ----------------------------------------------------------------------------------
    sync()                                   # ranks enter the first collective of the solve together
    solver.init(dev)
    costs = [solver.cost()]
    for _ in range(args.warmup):
        solver.step(dev)
        costs.append(solver.cost())
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.step(dev)
        if len(costs) < 3:
            costs.append(solver.cost())      # a host-side read of a stored scalar: no device work
    sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    cost_final = solver.cost()
    value = args.steps * args.liters / dt
    err = comm_error()
    if err:
        fail(err, "timed steps")

    gold, env = golden_cost(W, args.liters)
    parity = None
    if gold is not None and len(costs) >= 2:
        n = min(len(costs), len(gold))
        rel = [abs(a - b) / abs(b) for a, b in zip(costs[:n], gold[:n])]
        # Mid-trajectory costs after 400 float PCG iterations: the yardstick per step is how far apart LEGAL runs of the reference's own arithmetic end (its per-warp float
        # atomics commit in an undefined order; tools/reference_spread.py, profiles/r04_reference_order_spread.md), never below the 1e-5 contract.  Frozen at this size
        # where the runs exist (bench_<size>_float_400x2), else the spread of the 2048^2 one-step family at 400 iterations stands in (stated in `yardstick_source`).
        rs = reference_spread()
        key = f"bench_{W}_float_400x2"
        own = args.liters == 400 and rs.n_reference_order_runs(key) >= 2
        yard = [max(1e-5, (rs.spread(key, i) or 0.0) if i else 0.0) for i in range(n)] if own else [1e-5] + [rs.yardstick("horizon_2048_float_400", "float")] * (n - 1)
        parity = {"cost_hip": costs[:n], "cost_oracle_frozen": gold[:n], "rel_err": rel, "yardstick": yard, "factor": rs.FACTOR,
                  "yardstick_source": (f"diameter of {len(rs.legal_runs(key))} frozen legal runs of this workload (tests/golden/reference_order_costs.json {key}, bench_costs.json)" if own else
                                       "diameter of the frozen legal runs of horizon_2048_float_400 (one step, 2048^2): the 4096^2 runs are not frozen"),
                  "within_reference_spread": all(r <= rs.FACTOR * y for r, y in zip(rel, yard)),
                  "within_contract_1e-5": all(r <= 1e-5 for r in rel), "float_vs_double_oracle": env[:n] if env else None,
                  "source": "tests/golden/bench_costs.json (exact-order oracle float / double, tests/golden/make_bench_cost.py); tests/golden/reference_order_costs.json (make_reference_order_spread.py)"}

    # ---- roofline leg: the same plan goes on for two more steps with per-kernel hipEvents on the solver's stream -------------------
    roofline = None
    sha = kernel_src_sha16()
    comm_us = None
    if extra_steps:
        # single GPU: one event pair per RUN of launches of one name -- the 400 PCGIteration launches of a step cost two event records, not 800.  Row slabs: a pair per
        # launch (the communicator's kernels run between two iteration launches and must stay outside the brackets)
        solver.set_timing(1 if distributed else 2)
        if distributed:
            job.set_comm_timing(True)
        sync()
        tleg = time.perf_counter()
        for _ in range(extra_steps):
            solver.step(dev)
        sync()
        leg_ms_per_step = (time.perf_counter() - tleg) / extra_steps * 1e3     # this leg's own wall clock: the events it records cost time the timed steps do not pay
        kt = solver.kernel_timings()
        solver.set_timing(False)
        if distributed:
            ct = job.comm_timings()
            job.set_comm_timing(False)
            if ct:      # rank 0's hipEvents around the communicator's own kernels, per PCG iteration
                its = extra_steps * args.liters
                comm_us = {"allreduce_us_per_iteration": 1e3 * ct["allreduce"][1] / its, "allreduce_launches": ct["allreduce"][0],
                           "halo_us_per_iteration": 1e3 * ct["halo"][1] / its, "halo_exchanges": ct["halo"][0],
                           "note": "all-reduce = the one-workgroup kernel that sums this rank's partials and posts them to every mailbox (the wait for the peers' words "
                                   "happens in the next iteration kernel's prologue), or the waiting all-reduce; halo = push + pull kernels, one exchange per ghost - 1 iterations"}
        if rank == 0 and "PCGIteration" not in kt and "PCGSolveOnChip" in kt:
            # the slab fits the chip (8 x 4096x512): the whole linear solve is one persistent launch whose state never leaves registers / LDS -- HBM sees the loop's
            # inputs and delta once per Gauss-Newton step, so the kernel is bound by its grid-wide waits and VALU, not by a roofline of bytes
            cnt, tot = kt["PCGSolveOnChip"]
            rows = job.layout.rows if distributed else H
            per_step = {k: v[1] / extra_steps for k, v in kt.items()}
            bytes_per_launch = (24 + 4 + 1 + 12) * W * rows
            roofline = {"bound": "hbm", "kernel": "PCGSolveOnChip = iw_onchipPcg (the whole linear solve of a Gauss-Newton step in one persistent launch" + (", this rank's slab)" if distributed else ")"),
                        "achieved": bytes_per_launch / (tot / cnt * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_per_launch / (tot / cnt * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "note": "state on chip: HBM is touched at entry (r_0, p_0, angle, flags) and exit (delta) only; the launch is bound by one grid-wide wait per PCG iteration "
                                "plus VALU (DESIGN.md section 3.2), so the byte fraction is small by construction -- the figure of merit is us_per_iteration",
                        "avg_kernel_ms": tot / cnt, "launches": cnt, "us_per_iteration": 1e3 * tot / cnt / args.liters, "timed_leg_ms_per_step": leg_ms_per_step,
                        "streaming_equiv": {"bytes_per_pixel": MODEL_BYTES_PER_PIXEL["lattice"], "achieved": MODEL_BYTES_PER_PIXEL["lattice"] * W * rows * args.liters / (tot / cnt * 1e-3) / 1e9,
                                            "note": "what the streaming kernel would have to sustain to match (its 53 B/px per iteration over this launch's time); not a physical fraction"},
                        "kernel_ms_per_step": per_step, "kernel_avg_ms": {k: v[1] / v[0] for k, v in kt.items()}}
            if distributed:
                roofline.update({"slab_rows": rows, "ghost_rows": job.layout.ghost, "per_iteration_ms": dt / args.steps / args.liters * 1e3, "comm_kernels": comm_us})
        if rank == 0 and "PCGIteration" in kt:
            cnt, tot = kt["PCGIteration"]
            avg_s = tot / cnt * 1e-3
            rows = job.layout.rows if distributed else H
            model = MODEL_BYTES_PER_PIXEL["lattice"]
            achieved = model * W * rows / avg_s / 1e9
            traffic, traffic_src = measured_traffic(sha) if (W == 4096 and not distributed) else (None, None)
            hbm_achieved = traffic / avg_s / 1e9 if traffic else None
            algo = ALGO_BYTES_PER_PIXEL * W * rows / avg_s / 1e9
            per_step = {k: v[1] / extra_steps for k, v in kt.items()}
            roofline = {"bound": "hbm", "kernel": "PCGIteration = iw_pcgIter2 (PCGStep1+2+3 in one launch)" + (" on this rank's slab" if distributed else ""),
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": traffic_src, "hbm_achieved": hbm_achieved, "hbm_frac": hbm_achieved / HBM_PEAK_GBS if hbm_achieved else None,
                        "avg_kernel_ms": avg_s * 1e3, "launches": cnt, "model_bytes_per_pixel": model, "model_bytes_per_launch": model * W * rows,
                        "algorithmic_equiv": {"bytes_per_pixel": ALGO_BYTES_PER_PIXEL, "bytes_per_launch": ALGO_BYTES_PER_PIXEL * W * rows,
                                              "achieved": algo, "ratio_to_peak": algo / HBM_PEAK_GBS,
                                              "note": "reference formulation (3 kernels, SURVEY 8d) over this kernel's time; not a physical fraction"},
                        "timed_on": "the benchmarked plan itself: the two Opt_ProblemSteps after the timed ones, hipEvents switched on (OptAmd_PlanSetTiming mode 2: one event pair around each "
                                    "run of launches of one name -- the 400 PCGIteration launches of a step are bracketed ONCE, avg_kernel_ms = that interval / 400, launch gaps included)",
                        "kernel_ms_per_step": per_step, "kernel_ms_per_step_sum": sum(v for k, v in per_step.items() if k != "overall"),
                        "timed_leg_ms_per_step": leg_ms_per_step,
                        "timed_leg_note": "kernel_ms_per_step_sum <= timed_leg_ms_per_step (this leg's wall clock); with one event pair per run the leg costs what a timed step costs",
                        "kernel_avg_ms": {k: v[1] / v[0] for k, v in kt.items()}}
            if distributed:
                roofline.update({"slab_rows": rows, "ghost_rows": job.layout.ghost, "per_iteration_ms": dt / args.steps / args.liters * 1e3, "comm_kernels": comm_us,
                                 "note": "rank 0's slab; per_iteration_ms - avg_kernel_ms = communicator kernels (all-reduce every iteration, halo exchange every ghost - 1 "
                                         "iterations; launched by libOptComm, not in the table) + launch gaps"})

    # ---- the metric's own solve: Opt_ProblemSolve, 8 Gauss-Newton steps x 400 PCG iterations from the initial guess (main.cpp:113-114), timed as a whole ----
    solve = None
    if not args.no_extras:
        for slot in unknown_slots:
            dev[slot].copy_(torch.from_numpy(host0[slot]))
        solver.set_parameter("nIterations", 8)
        sync()
        t1 = time.perf_counter()
        solver.solve(dev)
        sync()
        sdt = max_over_ranks(time.perf_counter() - t1)
        final = solver.cost()
        gf, gd = golden_solve8(W) if args.liters == 400 else (None, None)
        solve = {"gn_solve_ms": sdt * 1e3, "what": f"one Opt_ProblemSolve (Init + 8 Steps x {args.liters} PCG iterations) from the initial guess, wall time incl. every host round trip, max over ranks",
                 "pcg_iters_per_s": 8 * args.liters / sdt, "final_energy": final}
        if gf:
            rel = abs(final - gf[-1]) / abs(gf[-1])
            envd = abs(gf[-1] - gd[-1]) / abs(gd[-1]) if gd else None
            # yardstick: the diameter of the frozen legal runs of THIS solve (exact-order plain / fma oracle, reference-order seeds where generated), never below the contract
            rs = reference_spread()
            key8 = f"solve8_{W}_float"
            y8 = rs.yardstick(key8, "float", 8)
            solve.update({"final_energy_oracle_float": gf[-1], "final_energy_oracle_double": gd[-1] if gd else None, "rel_err_vs_oracle_float": rel,
                          "oracle_float_vs_double": envd, "yardstick": y8, "factor": rs.FACTOR, "legal_runs": len(rs.legal_runs(key8, 8)), "reference_order_runs": rs.n_reference_order_runs(key8),
                          "within_contract_1e-5": rel <= 1e-5, "within_reference_spread": rel <= rs.FACTOR * y8,
                          "source": "tests/golden/horizon_costs.json / horizon_costs_fma.json / reference_order_costs.json solve8_* (oracle runs, generated offline)"})
        if comm_error():
            solve["comm_error"] = comm_error()
        # the reference's DEFAULT solver parameters (solverGPUGaussNewton.t:26-39: nIterations = 10, lIterations = 10): here the once-per-step kernels
        # (bind, PCGInit1, update, cost) weigh as much as the ten PCG launches between them
        for slot in unknown_slots:
            dev[slot].copy_(torch.from_numpy(host0[slot]))
        solver.set_parameter("nIterations", 10); solver.set_parameter("lIterations", 10)
        sync()
        t1 = time.perf_counter()
        solver.solve(dev)
        sync()
        ddt = max_over_ranks(time.perf_counter() - t1)
        solve["reference_default_10x10"] = {"solve_ms": ddt * 1e3, "ms_per_gn_step": ddt * 1e2, "pcg_iters_per_s": 100 / ddt, "final_energy": solver.cost()}
        solver.set_parameter("lIterations", args.liters)
        if not distributed:      # the same default shape at the reference's own image size (512^2: the linear solve runs on chip, the once-per-step launches weigh as much as it does)
            P5 = wl.image_warping(512, 512)
            d5 = api.to_device(P5)
            for kind in ("gaussNewtonGPU", "LMGPU"):
                s5 = api.Solver(api.energy_file("image_warping"), kind, (512, 512))
                s5.set_parameter("nIterations", 10); s5.set_parameter("lIterations", 10)
                for slot in P5.unknown_slots:
                    d5[slot].copy_(torch.from_numpy(P5.params[slot]))
                s5.solve(d5)      # warm-up (allocations, occupancy queries)
                for slot in P5.unknown_slots:
                    d5[slot].copy_(torch.from_numpy(P5.params[slot]))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                s5.solve(d5)
                torch.cuda.synchronize()
                t5 = time.perf_counter() - t1
                solve["reference_default_10x10_512_" + ("gn" if kind == "gaussNewtonGPU" else "lm")] = {"solve_ms": t5 * 1e3, "ms_per_outer_step": t5 * 1e2, "final_energy": s5.cost(), "on_chip_status": s5.on_chip_status()}
                s5.close()
            del d5

    # ---- the general kernel (arbitrary UrShape: + U, 61 B/pixel; M_a rebuilt from the pairs the stencil evaluates) on the same input ------
    general = None
    if not distributed and not args.no_extras:
        solver.close()
        del dev
        os.environ["OPT_AMD_LATTICE"] = "0"
        P3 = wl.image_warping(W, H)
        dev3 = api.to_device(P3)
        gs = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", (W, H))
        gs.set_parameter("nIterations", 4); gs.set_parameter("lIterations", args.liters)
        gs.init(dev3); gs.step(dev3)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            gs.step(dev3)
        torch.cuda.synchronize()
        gdt = time.perf_counter() - t1
        general = {"value": 3 * args.liters / gdt, "unit": "PCG iters/s", "model_bytes_per_pixel": MODEL_BYTES_PER_PIXEL["general"],
                   "note": "same workload with the unit-lattice specialisation switched off (OPT_AMD_LATTICE=0): the path any other UrShape takes"}
        gs.close()
        del dev3
        del os.environ["OPT_AMD_LATTICE"]

    # ---- the on-chip linear solve (opt_amd/csrc/iw_onchip.h) at the sizes it exists for, against the streaming loop on the same box ---------------------
    onchip = None
    flows = None
    if not distributed and not args.no_extras:
        onchip = onchip_table(api, wl, torch, args.liters)
        flows = reference_example_flows()
        if flows is not None:
            flows["parity_vs_frozen_reference_runs"] = flow_parity(flows)
    contract = None
    if not distributed and not args.no_extras:
        contract = contract_loop_leg(api, wl, torch, W, H, args.liters)
        contract["headline_over_contract_loop"] = value / contract["pcg_iters_per_s"]
    box = box_info(torch) if (rank == 0 and not args.share_gpu) else None
    if box and roofline and box.get("copy_gbs_float4"):
        best = max(box["copy_gbs_float4"], box["copy_gbs_float4_nontemporal"])
        roofline["frac_of_box_copy_float4"] = (roofline.get("hbm_achieved") or roofline["achieved"]) / best
        roofline["frac_of_box_copy_note"] = ("counter traffic (else model bytes) per launch / launch time over the better of this box's two float4 copy-kernel rates (box.copy_gbs_float4*): "
                                             "what a perfect streaming kernel reaches HERE")

    # ---- CPU leg last: the GPU work sits at the front of the run in one block ------------------------------------------------------------
    cpu = None
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_size, args.cpu_liters)

    # ---- the same slabs over RCCL (north_star names RCCL over xGMI): a short leg in the same job, so that RCCL demonstrably sees N ranks and both per-iteration
    # times stand side by side.  Ranks that share one GPU cannot form an RCCL communicator.
    rccl_leg = None
    if distributed and args.comm == "peer" and not args.no_extras:
        if args.share_gpu:
            rccl_leg = {"skipped": "needs distinct devices (ranks share one GPU in this run)"}
        else:
            err = comm_error()
            if err:
                fail(err, "before the RCCL leg")
            from opt_amd import slab
            job2 = slab.SlabJob("image_warping", W, H, rank, world, comm="rccl")
            job2.solver.set_parameter("nIterations", 3); job2.solver.set_parameter("lIterations", args.liters)
            sync()
            job2.solver.init(job2.params); job2.solver.step(job2.params)
            sync()
            t1 = time.perf_counter()
            job2.solver.step(job2.params); job2.solver.step(job2.params)
            sync()
            rdt = max_over_ranks(time.perf_counter() - t1)
            rccl_leg = {"comm": "rccl", "comm_ranks": job2.comm_ranks(), "steps": 2, "value": 2 * args.liters / rdt, "per_iteration_ms": rdt / 2 / args.liters * 1e3,
                        "cost_after_3_steps": job2.solver.cost(), "comm_error": job2.comm_error()}
            job2.close()

    if rank == 0:
        out = {"metric": f"PCG iters/s, GN solve of image_warping {W}^2", "value": value, "unit": "PCG iters/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"image_warping {W}x{H} float, gaussNewtonGPU, {args.liters} PCG iterations per GN step "
                                      "(synthetic cat512-style constraints, border pinned)",
                          "parallelism": f"row-slabs x{world}, comm={args.comm}, ranks in the communicator: {comm_ranks}" if distributed else "single GPU",
                          "step": "one Opt_ProblemStep (1 GN iteration)"},
               "gn_solve": solve, "gn_solve_ms": solve["gn_solve_ms"] if solve else None,
               "cost_initial": costs[0], "cost_final": cost_final, "parity": parity, "comm_ranks": comm_ranks, "preflight": preflight, "smoke": smoke, "rccl_leg": rccl_leg,
               "per_iteration_ms": dt / args.steps / args.liters * 1e3,
               "kernel_src_sha16": sha, "box": box, "roofline": roofline, "contract_loop": contract, "general_urshape": general, "onchip": onchip, "reference_example_flows": flows, "cpu_baseline": cpu}
        print(json.dumps(out))
    if distributed:
        job.close()
        dist.destroy_process_group()
    else:
        if general is None:
            solver.close()


if __name__ == "__main__":
    main()
