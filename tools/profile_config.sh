#!/bin/bash
# rocprofv3 kernel statistics + HBM traffic (separate PMC passes) of ONE BASELINE config run through tools/bench_configs.py:
#   tools/profile_config.sh <tag> <config substring>      e.g.  tools/profile_config.sh r01e_sfs config3
#   -> gpurun_out/<tag>/{kt, pmc_fetch, pmc_write};  then: python tools/summarize_config_profile.py gpurun_out/<tag> profiles/<tag>.md
tag=$1; cfg=$2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OPT_AMD_CONFIG="$cfg"
B="python tools/bench_configs.py"
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/kt -o kt -- $B > $out/kt.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -f csv -d $out/pmc_fetch -o p -- $B > $out/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -f csv -d $out/pmc_write -o p -- $B > $out/pmc_write.log 2>&1
ls $out/kt | head -3
