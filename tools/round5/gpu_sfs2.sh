#!/bin/bash
mkdir -p gpurun_out/r05s
timeout 600 python -u tools/round5/dbg_sfs_flow.py > gpurun_out/r05s/flow.txt 2>&1; echo "rc=$?" >> gpurun_out/r05s/flow.txt
cat gpurun_out/r05s/flow.txt
