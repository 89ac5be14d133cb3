#!/bin/bash
# One GPU-box call producing the artifacts a round commits under profiles/:
#   tools/round_artifacts.sh <tag>     ->  gpurun_out/<tag>/{kt/, pmc_fetch/, pmc_write/, bench.json, configs.json}
# then, back in the container:  python tools/summarize_profile.py gpurun_out/<tag> profiles/<tag>
# Every rocprofv3 run sits under `timeout`; counters are collected in their own passes (never with trace options).
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# clocks and power cap of this box (the 10 % box-to-box spread of the bench line: sclk / mclk / cap differ between boxes), before and after the runs
smi() { { date -u +%FT%TZ; rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showtemp 2>&1 | grep -v "^$\|====" ; } >> $out/rocm_smi.txt 2>&1; }
smi
B="python bench.py --steps 1 --warmup 0 --liters 50 --no-cpu-baseline --no-extras"
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/kt -o kt -- $B > $out/kt.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -f csv -d $out/pmc_fetch -o p -- $B > $out/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -f csv -d $out/pmc_write -o p -- $B > $out/pmc_write.log 2>&1
# the on-chip linear solve (one persistent launch per Gauss-Newton step) at the sizes it exists for: kernel trace of 4096x512 and 512x512, 400 iterations per launch
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/kt_onchip -o kt -- python tools/onchip_bench.py --sizes 4096x512,512x512 --steps 4 > $out/kt_onchip.log 2>&1
cp $out/kt_onchip/*kernel_stats.csv $out/onchip_kernel_stats.csv 2>/dev/null
# The counters are folded into profiles/<tag>_traffic.json on the box first, so that the bench line below carries the traffic measured for this very tree
# (bench.py accepts a traffic file only if its kernel-source hash matches).
python tools/summarize_profile.py $out profiles/$tag > $out/summarize.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
timeout 600 python tools/bench_configs.py > $out/configs.json 2> $out/configs.err
smi
cp $out/rocm_smi.txt profiles/${tag}_rocm_smi.txt 2>/dev/null
tail -c 600 $out/bench.json
