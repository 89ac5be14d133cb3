#!/bin/bash
# Grid-wide sum + halo hand-over latencies for a would-be persistent slab kernel (tools/microbench_gridsync.hip)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03v
hipcc --offload-arch=gfx950 -O3 -o /tmp/gridsync tools/microbench_gridsync.hip || exit 1
timeout 120 /tmp/gridsync 2000 > gpurun_out/r03v/gridsync.txt 2>&1; echo "rc=$?" >> gpurun_out/r03v/gridsync.txt
cat gpurun_out/r03v/gridsync.txt
