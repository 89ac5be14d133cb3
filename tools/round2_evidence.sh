#!/bin/bash
# One GPU-box call that re-measures what DESIGN.md quotes for round 2 and leaves it under gpurun_out/<tag>/ (copied to profiles/<tag>_*.txt afterwards).
tag=${1:-r02c}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rate() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f PCG it/s  (%.1f us per iteration)' % (d['value'], 1e6/d['value']))"; }
{
  echo "# memory skeleton of the iteration kernel (tools/microbench_seq.hip, r-free byte pattern) and the real kernel on the same box"
  timeout 300 ./tools/mb_seq geo 2>&1 | head -17
  echo -n "real kernel (bench.py, 3 steps x 400 iterations): "; $B 2>/dev/null | rate
} > $out/skeleton_vs_kernel.txt 2>&1
{
  echo "# interleaved A/B of the byte-saving reformulations (bench.py --steps 3, one box)"
  for r in 1 2 3; do
    for e in "OPT_AMD_RFREE=1" "OPT_AMD_RFREE=0 OPT_AMD_RECON_P=1" "OPT_AMD_RFREE=0 OPT_AMD_RECON_P=0" "OPT_AMD_LATTICE=0"; do echo -n "$e: "; env $e $B 2>/dev/null | rate; done
  done
} > $out/ab_bytes.txt 2>&1
{
  echo "# plain (default) vs non-temporal loads (libOpt_nt1.so = -DIW_NT_LOAD=1) over image sizes, interleaved"
  bash tools/size_ab.sh libOpt.so libOpt_nt1.so 2>&1 | grep us/iter
} > $out/size_ab_nt.txt 2>&1
{
  echo "# per-iteration cost of the communication path on one GPU (1-rank slab job, communicator forced on)"
  timeout 500 python tools/slab_overhead.py 2>&1 | grep "us per"
  echo "# kernels of the 4096x512 slab job with the peer communicator (rocprofv3 --kernel-trace --stats)"
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/slabprof -o kt -- python tools/slab_profile.py peer > $out/slabprof.log 2>&1
  head -6 $out/slabprof/kt_kernel_stats.csv | cut -c1-260
} > $out/slab_overhead.txt 2>&1
{
  echo "# config 3 (SFS 1024^2 double LM 60x10): one kernel per PCG iteration vs three, interleaved"
  bash tools/ab_sfs.sh "OPT_AMD_SFS_ONEKERNEL=1" "OPT_AMD_SFS_ONEKERNEL=0" 2>&1
} > $out/ab_sfs.txt 2>&1
{ echo "# dependent-launch and round-trip costs"; ./tools/mb_launch; } > $out/mb_launch.txt 2>&1
{ echo "# cost parity against the float oracle at short horizons (relative difference)"; python tools/parity_table.py 2>/dev/null; } > $out/parity_table.md 2>&1
{ echo "# 2 ranks as processes sharing one GPU through the whole bench path (functional check; timings meaningless)"; timeout 300 python bench.py --gpus 2 --share-gpu --size 2048 --steps 2 --warmup 1 2>/dev/null | tail -1; } > $out/bench_2ranks_shared_gpu.json 2>&1
tail -n +1 $out/*.txt $out/parity_table.md | head -150
