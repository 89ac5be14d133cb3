#!/bin/bash
# round 5: phase profile of the on-chip shape_from_shading kernel (development variant libOpt_sfsprof.so)
mkdir -p gpurun_out/r05s
[ -f opt_amd/lib/libOpt_sfsprof.so ] || python -c 'from opt_amd import build; build.build_variant("sfsprof", ["SO_PROFILE=1"])' > /dev/null      # (the development variant is not kept in the tree)
export OPT_AMD_LIB=$PWD/opt_amd/lib/libOpt_sfsprof.so OPT_AMD_ONCHIP_PROFILE=1
timeout 300 python -u - > gpurun_out/r05s/prof.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from opt_amd import api, workloads as wl
for (W, H, dbl) in [(640, 480, True), (1024, 1024, True), (640, 480, False)]:
    P = wl.shape_from_shading(W, H, double=dbl, seed=1, holes=True)
    for rows, waves in ((None, None), (4, 8), (8, 4)):
        for k, v in (("OPT_AMD_ONCHIP_ROWS", rows), ("OPT_AMD_ONCHIP_WAVES", waves)):
            if v: os.environ[k] = str(v)
            else: os.environ.pop(k, None)
        g = api.Solver(api.energy_file(P.energy), "LMGPU", P.dims, double=dbl)
        g.set_parameter("nIterations", 3); g.set_parameter("lIterations", 10)
        dev = api.to_device(P)
        g.init(dev)
        for _ in range(3): g.step(dev)
        g.close()
PY
echo "rc=$?" >> gpurun_out/r05s/prof.txt
grep "profile\|rc=" gpurun_out/r05s/prof.txt
