#!/bin/bash
# kernel timeline of one config's plain solve:  tools/timeline.sh <tag> <config substring>  -> gpurun_out/<tag>/gaps.txt
tag=$1; cfg=$2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OPT_AMD_CONFIG="$cfg" OPT_AMD_NO_TIMING_RUN=1
timeout 200 rocprofv3 --kernel-trace -f csv -d $out/kt -o kt -- python tools/bench_configs.py > $out/kt.log 2>&1
python tools/timeline_gaps.py $out/kt > $out/gaps.txt 2>&1
cat $out/gaps.txt
