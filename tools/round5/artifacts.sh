#!/bin/bash
# round 5: everything profiles/r05_* is made of, in ONE GPU-box call (every rocprofv3 run under `timeout`, counters in their own passes)
tag=${1:-r05}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
bash tools/round_artifacts.sh $tag > $out/round_artifacts.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) + kernel trace of configs 1 (+ poisson 2048^2), 3 and 4 and of the on-chip kernel
for cfg in "config1" "poisson_image_editing 2048" config3 config4; do
  d=$out/pmc_$(echo $cfg | cut -d' ' -f1)
  mkdir -p $d
  export OPT_AMD_CONFIG="$cfg" OPT_AMD_NO_TIMING_RUN=1
  B="python tools/bench_configs.py"
  timeout 150 rocprofv3 --kernel-trace --stats -f csv -d $d/kt -o kt -- $B > $d/kt.log 2>&1
  timeout 150 rocprofv3 --pmc FETCH_SIZE -f csv -d $d/pmc_fetch -o p -- $B > $d/pmc_fetch.log 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE -f csv -d $d/pmc_write -o p -- $B > $d/pmc_write.log 2>&1
  unset OPT_AMD_CONFIG OPT_AMD_NO_TIMING_RUN
done
d=$out/pmc_onchip; mkdir -p $d
B="python tools/onchip_bench.py --sizes 4096x512,512x512 --steps 3"
timeout 150 rocprofv3 --pmc FETCH_SIZE -f csv -d $d/pmc_fetch -o p -- $B > $d/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE -f csv -d $d/pmc_write -o p -- $B > $d/pmc_write.log 2>&1
python tools/config_rooflines.py $out profiles/${tag}_configs.json > $out/config_rooflines.txt 2>&1
# what a multi-GPU run WOULD do, per rank (single GPU, and 8 ranks sharing this GPU at the metric's size and at config 5's)
python bench.py --dry > $out/dry_1.json 2> $out/dry_1.err
timeout 300 python bench.py --gpus 8 --share-gpu --dry > $out/dry_8_4096.json 2> $out/dry_8_4096.err
timeout 300 python bench.py --gpus 2 --share-gpu --dry > $out/dry_2_4096.json 2> $out/dry_2_4096.err
# the long-horizon tables
timeout 600 python tools/horizon_parity.py --out $out/horizon_parity > $out/horizon_parity.log 2>&1
cat $out/config_rooflines.txt; tail -c 400 $out/dry_8_4096.json
