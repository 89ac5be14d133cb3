"""poisson_image_editing 2048x2048 float GN 1x100 (bench_configs' plumbing config at size): wall time and PCGIteration kernel time under OPT_AMD_ITER_ROWS values."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from opt_amd import api, workloads as wl

def run(rows, W=2048, H=2048, lit=100, double=False):
    if rows: os.environ["OPT_AMD_ITER_ROWS"] = str(rows)
    else: os.environ.pop("OPT_AMD_ITER_ROWS", None)
    P = wl.poisson_image_editing(W, H, double=double)
    best = 1e9
    for timing in (False, False, False, True):
        g = api.Solver(api.energy_file(P.energy), "gaussNewtonGPU", P.dims, double=double, timing=timing)
        g.set_parameter("nIterations", 1); g.set_parameter("lIterations", lit)
        Q = P.clone(); dev = api.to_device(Q)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.solve(dev)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if not timing: best = min(best, dt)
        else: kt = g.kernel_timings()
        c = g.cost(); g.close()
    return best * 1e3, kt["PCGIteration"][1] / kt["PCGIteration"][0] * 1e3, c

if __name__ == "__main__":
    for rows in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,12,24,36,48,64,96,128").split(",")]:
        ms, us, c = run(rows)
        print(f"rows {rows:4d}: solve {ms:.3f} ms, PCGIteration {us:.1f} us, cost {c:.6g}", flush=True)
    if os.environ.get("SWEEP_H"):
        for H in (128, 256, 512, 1024, 2048, 4096):
            ms, us, c = run(0, 2048, H)
            print(f"2048x{H}: solve {ms:.3f} ms, PCGIteration {us:.1f} us  ({65 * 2048 * H / us / 1e6:.2f} TB/s of 65 B/px)", flush=True)
