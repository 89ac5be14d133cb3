#!/bin/bash
# One GPU-box call producing the artifacts a round commits under profiles/:
#   tools/round_artifacts.sh <tag>     ->  gpurun_out/<tag>/{kt/, pmc_fetch/, pmc_write/, bench.json, configs.json}
# then, back in the container:  python tools/summarize_profile.py gpurun_out/<tag> profiles/<tag>
# Every rocprofv3 run sits under `timeout`; counters are collected in their own passes (never with trace options).
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 1 --warmup 0 --liters 50 --no-cpu-baseline --no-extras"
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/kt -o kt -- $B > $out/kt.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -f csv -d $out/pmc_fetch -o p -- $B > $out/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -f csv -d $out/pmc_write -o p -- $B > $out/pmc_write.log 2>&1
# The counters are folded into profiles/<tag>_traffic.json on the box first, so that the bench line below carries the traffic measured for this very tree
# (bench.py accepts a traffic file only if its kernel-source hash matches).
python tools/summarize_profile.py $out profiles/$tag > $out/summarize.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
timeout 600 python tools/bench_configs.py > $out/configs.json 2> $out/configs.err
tail -c 600 $out/bench.json
