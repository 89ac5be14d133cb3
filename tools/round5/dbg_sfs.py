"""round 5: first light of the on-chip shape_from_shading solve -- cost after one outer step, on chip against the marching kernels, per variant / kind / iterations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from opt_amd import api, workloads as wl
import torch


def run(P, kind, L, flag, rows=None, steps=1, waves=None):
    os.environ["OPT_AMD_ONCHIP"] = flag
    if rows: os.environ["OPT_AMD_ONCHIP_ROWS"] = str(rows)
    else: os.environ.pop("OPT_AMD_ONCHIP_ROWS", None)
    if waves: os.environ["OPT_AMD_ONCHIP_WAVES"] = str(waves)
    else: os.environ.pop("OPT_AMD_ONCHIP_WAVES", None)
    g = api.Solver(api.energy_file(P.energy), kind, P.dims, double=P.double, timing=True)
    g.set_parameter("nIterations", steps); g.set_parameter("lIterations", L)
    dev = api.to_device(P)
    g.init(dev)
    c = [g.cost()]
    for _ in range(steps):
        g.step(dev); c.append(g.cost())
    t = g.kernel_timings()
    st = g.on_chip_status()
    x = torch.cat([dev[i].reshape(-1) for i in P.unknown_slots]).cpu().numpy()
    g.close()
    return c, st, x, t


for (W, H) in [(40, 32), (130, 37), (200, 150)]:
    for dbl in (True, False):
        P = wl.shape_from_shading(W, H, double=dbl, seed=3, holes=True, noise=2e-3)
        for kind in ("gaussNewtonGPU", "LMGPU"):
            for L in (1, 2, 10):
                c0, _, x0, _ = run(P, kind, L, "0")
                for rows, waves in ((4, 4), (6, 8), (8, 4), (10, 8)):
                    c1, st, x1, t = run(P, kind, L, "1", rows, waves=waves)
                    rel = abs(c1[1] - c0[1]) / abs(c0[1])
                    dx = np.linalg.norm(x1 - x0) / max(np.linalg.norm(x0), 1e-300)
                    print(f"{W}x{H} {'f64' if dbl else 'f32'} {kind[:2]} L={L} R={rows} status={st} cost {c0[0]:.6g} -> march {c0[1]:.12g} onchip {c1[1]:.12g} rel={rel:.2e} dx={dx:.2e} "
                          f"{'OK' if rel < (1e-9 if dbl else 1e-5) and st == 1 else 'BAD'}", flush=True)

# timing at the reference's input size and at config 3
for (W, H, steps) in [(640, 480, 5), (1024, 1024, 5)]:
    P = wl.shape_from_shading(W, H, double=True, seed=1, holes=True)
    for flag, rows, waves in (("0", None, None), ("1", None, None), ("1", 4, 8), ("1", 6, 4), ("1", 6, 8), ("1", 8, 4), ("1", 8, 8), ("1", 10, 4), ("1", 10, 8)):
        c, st, x, t = run(P, "LMGPU", 10, flag, rows, steps=steps, waves=waves)
        ks = {k: (v[0], round(1e3 * v[1] / max(v[0], 1), 2)) for k, v in t.items() if "PCG" in k}
        print(f"{W}x{H} f64 LM flag={flag} R={rows} waves={waves} status={st} costs {c[0]:.8g} -> {c[-1]:.12g}  kernels (count, us avg): {ks}", flush=True)
for (W, H, steps) in [(640, 480, 5), (512, 512, 5)]:
    P = wl.shape_from_shading(W, H, double=False, seed=1, holes=True)
    for flag, rows, waves in (("0", None, None), ("1", None, None), ("1", 4, 8), ("1", 6, 4), ("1", 8, 4)):
        c, st, x, t = run(P, "LMGPU", 10, flag, rows, steps=steps, waves=waves)
        ks = {k: (v[0], round(1e3 * v[1] / max(v[0], 1), 2)) for k, v in t.items() if "PCG" in k}
        print(f"{W}x{H} f32 LM flag={flag} R={rows} waves={waves} status={st} costs {c[0]:.8g} -> {c[-1]:.12g}  kernels (count, us avg): {ks}", flush=True)
