#!/bin/bash
# memory skeleton of the iteration kernel: row-major vectors against strip-major ones (each workgroup's streams contiguous)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03ai
hipcc --offload-arch=gfx950 -O3 -DRFREE -DRECON_P -o /tmp/mbseq_row tools/microbench_seq.hip || exit 1
hipcc --offload-arch=gfx950 -O3 -DRFREE -DRECON_P -DSTRIPMAJOR -o /tmp/mbseq_strip tools/microbench_seq.hip || exit 1
{ echo "== row-major"; timeout 120 /tmp/mbseq_row s; echo "== strip-major"; timeout 120 /tmp/mbseq_strip s; echo "== row-major"; timeout 120 /tmp/mbseq_row s; } 2>&1 | tee gpurun_out/r03ai/skeleton_stripmajor.txt
