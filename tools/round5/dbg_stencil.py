"""round 5: on-chip against marching kernels for the 5-point stencil energies -- kernel time per linear solve and wall time of a whole solve."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from opt_amd import api, workloads as wl
import torch

CASES = [("poisson 256^2 f32 GN 1x10 (config 1)", lambda: wl.poisson_image_editing(256, 256, seed=1), 1, 10),
         ("poisson 512^2 f32 GN 1x100", lambda: wl.poisson_image_editing(512, 512, seed=1), 1, 100),
         ("poisson 700x700 f32 GN 1x100", lambda: wl.poisson_image_editing(700, 700, seed=1), 1, 100),
         ("poisson 1000x1000 f32 GN 1x100", lambda: wl.poisson_image_editing(1000, 1000, seed=1), 1, 100),
         ("laplacian 512^2 f32 GN 1x50", lambda: wl.laplacian(512, 512, seed=1), 1, 50),
         ("optical_flow 512^2 f32 GN 3x50", lambda: wl.optical_flow(512, 512, seed=1), 3, 50),
         ("optical_flow 960x540 f32 GN 3x50", lambda: wl.optical_flow(960, 540, seed=1), 3, 50),
         ("optical_flow 1024^2 f32 GN 3x50", lambda: wl.optical_flow(1024, 1024, seed=1), 3, 50),
         ("intrinsic 512^2 f32 GN 7x10", lambda: wl.intrinsic_image_decomposition(512, 512, seed=1), 7, 10),
         ("intrinsic 1024^2 f32 GN 7x10", lambda: wl.intrinsic_image_decomposition(1024, 1024, seed=1), 7, 10)]
CASES = [c + ("gaussNewtonGPU",) for c in CASES] + [("poisson 512^2 f32 LM 5x10", lambda: wl.poisson_image_editing(512, 512, seed=1), 5, 10, "LMGPU"),
                                                   ("optical_flow 512^2 f32 LM 5x10", lambda: wl.optical_flow(512, 512, seed=1), 5, 10, "LMGPU"),
                                                   ("laplacian 512^2 f32 LM 5x10", lambda: wl.laplacian(512, 512, seed=1), 5, 10, "LMGPU")]
for name, make, nit, lit, kind in CASES:
    for flag in ("0", "1"):
        os.environ["OPT_AMD_ONCHIP"] = flag
        out = []
        for timing in (False, True):
            P = make()
            g = api.Solver(api.energy_file(P.energy), kind, P.dims, double=P.double, timing=timing)
            g.set_parameter("nIterations", nit); g.set_parameter("lIterations", lit)
            dev = api.to_device(P); g.solve(dev)
            dev = api.to_device(P)
            torch.cuda.synchronize(); t0 = time.perf_counter(); g.solve(dev); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if timing:
                t = g.kernel_timings()
                out.append(", ".join(f"{k} {v[0]}x{1e3 * v[1] / max(v[0], 1):.1f}us" for k, v in t.items() if k.startswith("PCG")))
            else:
                out.append(f"wall {dt * 1e3:.3f} ms cost {g.cost():.8g} status {g.on_chip_status()} {g.describe().get('onchip_rows_per_wave', '-')}/{g.describe().get('waves_per_workgroup', '-')}")
            g.close()
        print(f"{name} onchip={flag}: {out[0]} | {out[1]}", flush=True)
