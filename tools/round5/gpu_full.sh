#!/bin/bash
mkdir -p gpurun_out/r05e
timeout 1500 python -m pytest tests -q -m gpu --maxfail=10 > gpurun_out/r05e/full.txt 2>&1; echo "rc=$?" >> gpurun_out/r05e/full.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/r05e/smoke.txt 2>&1; echo "rc=$?" >> gpurun_out/r05e/smoke.txt
grep -v "^E  \|^$" gpurun_out/r05e/full.txt | tail -n 30; tail -3 gpurun_out/r05e/smoke.txt
