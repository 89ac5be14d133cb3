#!/bin/bash
mkdir -p gpurun_out/r05t
timeout 600 python -u tools/round5/dbg_stencil.py > gpurun_out/r05t/timing.txt 2>&1; echo "rc=$?" >> gpurun_out/r05t/timing.txt
grep -v amdgpu.ids gpurun_out/r05t/timing.txt
