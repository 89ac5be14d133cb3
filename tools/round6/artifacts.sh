#!/bin/bash
# round 6: everything profiles/r06_* is made of, in ONE GPU-box call (every rocprofv3 run under `timeout`, counters in their own passes).  The raw per-dispatch CSVs are
# folded into tables ON THE BOX and deleted (gpurun brings back 64 MiB at most); what travels back is gpurun_out/<tag>/{*.json, *.md, *.csv summaries, *.txt}.
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
bash tools/round_artifacts.sh $tag > $out/round_artifacts.log 2>&1
cp profiles/${tag}_* $out/ 2>/dev/null      # (written on the box by summarize_profile.py: the traffic file the bench line of this tree refers to, the kernel statistics, the summary)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
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
python tools/config_rooflines.py $out $out/${tag}_configs.json > $out/config_rooflines.txt 2>&1
# kernel statistics of the per-config traces, then the raw rows go
for cfg in config1 poisson_image_editing config3 config4; do cp $out/pmc_$cfg/kt/*/*kernel_stats.csv $out/${tag}_${cfg}_kernel_stats.csv 2>/dev/null || cp $out/pmc_$cfg/kt/*kernel_stats.csv $out/${tag}_${cfg}_kernel_stats.csv 2>/dev/null; done
rm -rf $out/pmc_* $out/kt $out/kt_onchip $out/pmc_fetch $out/pmc_write
# what a multi-GPU run WOULD do, per rank (single GPU, and 2 / 8 ranks sharing this GPU at the metric's size)
python bench.py --dry > $out/dry_1.json 2> $out/dry_1.err
timeout 300 python bench.py --gpus 8 --share-gpu --dry > $out/dry_8_4096.json 2> $out/dry_8_4096.err
timeout 300 python bench.py --gpus 2 --share-gpu --dry > $out/dry_2_4096.json 2> $out/dry_2_4096.err
# the N-rank path end to end on one GPU (smoke solve, posted all-reduce, on-chip slab solve): functional, timings meaningless
timeout 600 python bench.py --gpus 8 --share-gpu --size 4096 --steps 1 --warmup 0 --liters 40 --no-cpu-baseline > $out/bench_8ranks_shared_gpu_4096.json 2> $out/bench_8ranks.err
# one-GPU slab plumbing table and the long-horizon tables
timeout 600 python tools/slab_overhead.py > $out/slab_overhead.txt 2>&1
timeout 600 python tools/horizon_parity.py --out $out/horizon_parity > $out/horizon_parity.log 2>&1
timeout 300 python tools/config_horizon.py --out $out/config_horizon.json > $out/config_horizon.txt 2>&1
du -sh $out; cat $out/config_rooflines.txt | tail -20; tail -c 400 $out/dry_8_4096.json; tail -n 12 $out/slab_overhead.txt; cat $out/config_horizon.txt
