#!/bin/bash
mkdir -p gpurun_out/r05n
timeout 120 python tools/slab_overhead.py > gpurun_out/r05n/slab_overhead.txt 2> gpurun_out/r05n/slab_overhead.err; echo "rc=$?" >> gpurun_out/r05n/slab_overhead.txt
cat gpurun_out/r05n/slab_overhead.txt | cut -c1-200
