#!/usr/bin/env python
"""One short 1-rank slab solve with the peer communicator forced on (for rocprofv3 --kernel-trace): 4096x512, 2 steps x 200 iterations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from opt_amd import slab
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29578", OPT_AMD_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
mode = sys.argv[1] if len(sys.argv) > 1 else "peer"
job = slab.SlabJob("image_warping", 4096, 512, 0, 1, comm=mode)
s = job.solver
s.set_parameter("nIterations", 2); s.set_parameter("lIterations", 200)
s.init(job.params)
while s.step(job.params): pass
torch.cuda.synchronize()
job.close()
dist.destroy_process_group()
