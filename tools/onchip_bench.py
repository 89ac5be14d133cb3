"""PCG iterations per second of image_warping at the sizes the on-chip linear solve (opt_amd/csrc/iw_onchip.h) exists for, against the streaming
loop (one launch per PCG iteration) on the same box: the reference's own inputs (512^2: examples/image_warping/src/main.cpp:98-134; the SFS fixture's
640x480), 1024^2, 2048x1024 and 4096x512 (1/8 of the metric's 4096^2).  Writes a markdown table to stdout.

    python tools/onchip_bench.py [--liters 400] [--steps 6] [--sizes 512x512,640x480,...] [--env KEY=VALUE ...]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(W, H, liters, steps, onchip, double=False):
    import torch
    from opt_amd import api, workloads as wl
    os.environ["OPT_AMD_ONCHIP"] = "1" if onchip else "0"
    P = wl.image_warping(W, H, double=double)
    g = api.Solver(api.energy_file(P.energy), "gaussNewtonGPU", P.dims, double=double, timing=False)
    g.set_parameter("nIterations", steps + 2); g.set_parameter("lIterations", liters)
    dev = api.to_device(P)
    g.init(dev)
    g.step(dev); g.step(dev)                      # warm-up (allocations, occupancy queries, first-touch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.step(dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cost = g.cost()
    # what ran: a timed plan of two steps
    g2 = api.Solver(api.energy_file(P.energy), "gaussNewtonGPU", P.dims, double=double, timing=True)
    g2.set_parameter("nIterations", 2); g2.set_parameter("lIterations", liters)
    dev2 = api.to_device(P)
    g2.solve(dev2)
    kt = g2.kernel_timings()
    g.close(); g2.close()
    return dt / steps, cost, kt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--liters", type=int, default=400)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--sizes", default="512x512,640x480,1024x1024,2048x1024,4096x512")
    ap.add_argument("--double", action="store_true")
    a = ap.parse_args()
    print(f"| image | pixels | on-chip: us per PCG iteration | PCG it/s | kernel us per iteration (hipEvents) | streaming: us per PCG iteration | PCG it/s | speed-up | cost after {a.steps + 2} steps (on-chip / streaming) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for s in a.sizes.split(","):
        W, H = (int(v) for v in s.split("x"))
        t_on, c_on, k_on = run(W, H, a.liters, a.steps, True, a.double)
        t_st, c_st, k_st = run(W, H, a.liters, a.steps, False, a.double)
        oc = k_on.get("PCGSolveOnChip")
        kus = f"{1e3 * oc[1] / oc[0] / a.liters:.2f}" if oc else "(not taken)"
        print(f"| {W}x{H} | {W * H} | {1e6 * t_on / a.liters:.2f} | {a.liters / t_on:.0f} | {kus} | {1e6 * t_st / a.liters:.2f} | {a.liters / t_st:.0f} | {t_st / t_on:.2f}x | {c_on:.6g} / {c_st:.6g} |", flush=True)


if __name__ == "__main__":
    main()
