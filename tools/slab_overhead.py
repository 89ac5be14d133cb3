#!/usr/bin/env python
"""Per-iteration overhead of the multi-GPU plumbing, measurable on ONE GPU: the same slab-sized image_warping problem solved (a) as a plain
single-GPU problem, (b) as a 1-rank slab job with the peer-mailbox communicator forced on (every all-reduce / halo kernel runs, peers =
self), (c) the same with RCCL.  The difference (a) -> (b)/(c) is what the communication path adds per PCG iteration before any xGMI latency.
Modes: plain (on-chip where the image fits), onchip-slab (the on-chip solve as a 1-rank slab job: decision all-reduce, ghost-row exchange, the rank hop of the sums through
the mailbox), plain-streaming / peer-post / peer-wait / rccl (one launch per PCG iteration, OPT_AMD_ONCHIP=0)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(mode, W, H, liters, steps):
    import torch
    import torch.distributed as dist
    from opt_amd import api, slab, workloads as wl
    os.environ["OPT_AMD_ONCHIP"] = "0" if mode.endswith("-streaming") or mode in ("peer", "peer-post", "peer-wait", "rccl") else "1"
    if mode == "onchip-slab":
        os.environ["OPT_AMD_PEER_POST"] = "1"; os.environ["OPT_AMD_PEER_PLAN"] = "0"
        job = slab.SlabJob("image_warping", W, H, 0, 1, comm="peer")
        s, dev = job.solver, job.params
    elif mode in ("plain", "plain-streaming"):
        P = wl.image_warping(W, H)
        dev = api.to_device(P)
        s = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", (W, H))
        job = None
    else:
        # "peer": the all-reduce is posted and polled by the next iteration kernel's prologue (round 3); "peer-wait": a kernel between two launches waits for it (round 2)
        os.environ["OPT_AMD_PEER_POST"] = "0" if mode == "peer-wait" else "1"
        os.environ["OPT_AMD_PEER_PLAN"] = "1" if mode == "peer" else "0"          # "peer": the iteration kernel's last workgroup posts; "peer-post": a one-workgroup kernel posts
        job = slab.SlabJob("image_warping", W, H, 0, 1, comm="peer" if mode.startswith("peer") else mode)
        s, dev = job.solver, job.params
    s.set_parameter("nIterations", steps + 1); s.set_parameter("lIterations", liters)
    s.init(dev); s.step(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step(dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if job:
        job.close()
    else:
        s.close()
    return dt / (steps * liters) * 1e6


def main():
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", OPT_AMD_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    out = {}
    sizes = [(4096, 512), (4096, 1024), (4096, 2048), (8192, 1024)] if "--all" in sys.argv else [(4096, 512), (8192, 1024)]
    for (W, H) in sizes:
        for mode in (("plain-streaming", "peer", "peer-post") if "--plan" in sys.argv else ("plain", "onchip-slab", "plain-streaming", "peer-post", "peer-wait", "rccl")):
            us = run(mode, W, H, 400, 3)
            out[f"{W}x{H}_{mode}"] = us
            print(f"{W}x{H:5d} {mode:16s}: {us:7.1f} us per PCG iteration", flush=True)
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
