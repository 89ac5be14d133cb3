#!/bin/bash
# arithmetic of a would-be on-chip slab kernel per iteration, by where the state lives (tools/microbench_onchip.hip)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03aw
for v in "-DCS_REGS=1 -DAP_LDS=0 -DDELTA_LDS=1" "-DCS_REGS=0 -DAP_LDS=0 -DDELTA_LDS=1" "-DCS_REGS=1 -DAP_LDS=1 -DDELTA_LDS=0" "-DCS_REGS=0 -DAP_LDS=1 -DDELTA_LDS=0" "-DCS_REGS=1 -DAP_LDS=1 -DDELTA_LDS=0 -DDELTA_ATOMIC=1" "-DCS_REGS=0 -DAP_LDS=1 -DDELTA_LDS=0 -DDELTA_ATOMIC=1"; do
  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize $v -o /tmp/onchip tools/microbench_onchip.hip || exit 1
  timeout 120 /tmp/onchip 200
done 2>&1 | tee gpurun_out/r03aw/onchip.txt
