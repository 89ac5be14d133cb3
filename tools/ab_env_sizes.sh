#!/bin/bash
# tools/ab_env_sizes.sh "ENV=.." "ENV=.." ... : per-iteration time of the plain single-GPU image_warping solve over sizes, interleaved over environment settings
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for e in "$@"; do
env $e python - <<PY
import time, torch
from opt_amd import api, workloads as wl
for (W,H) in [(4096,512),(4096,1024),(2048,2048),(4096,2048),(4096,4096)]:
    P = wl.image_warping(W,H); dev = api.to_device(P)
    s = api.Solver(api.energy_file("image_warping"), "gaussNewtonGPU", (W,H))
    s.set_parameter("nIterations", 4); s.set_parameter("lIterations", 400)
    s.init(dev); s.step(dev); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(3): s.step(dev)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("$e %dx%d: %.1f us/iter  cost %.9g" % (W,H,dt/1200*1e6, s.cost()), flush=True)
    s.close()
PY
done
done
