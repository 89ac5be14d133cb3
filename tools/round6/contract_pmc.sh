#!/bin/bash
# HBM traffic of the reference-ordered loop's kernels at 4096^2 (FETCH_SIZE / WRITE_SIZE in their own passes + a kernel trace): is the "149 B/px physically moved" of
# bench.py's contract_loop.roofline a measurement?
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python tools/round6/contract_loop_run.py"
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/kt -o kt -- $B > $out/kt.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -f csv -d $out/p1 -o p -- $B > $out/p1.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -f csv -d $out/p2 -o p -- $B > $out/p2.log 2>&1
python tools/round6/pmc_by_kernel.py $out > $out/pmc_by_kernel.txt 2>&1
rm -rf $out/p1 $out/p2 $out/kt
grep -A3 "k_step2\|iw_applyJTJ\|k_step3" $out/pmc_by_kernel.txt | head -40
