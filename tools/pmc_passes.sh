#!/bin/bash
# Collect rocprofv3 PMC passes (each in its own run, as gpurun requires) for a bench command.
#   tools/pmc_passes.sh <outdir> [short] <env assignments...>
# Every pass is wrapped in `timeout`: a rocprofv3 counter pass that aborts can otherwise hang until gpurun's limit.
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
B="python bench.py --steps 1 --warmup 0 --liters 20 --no-cpu-baseline --no-extras"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ"; do
  i=$((i+1))
  env "$@" timeout 120 rocprofv3 --pmc $set -f csv -d $out/p$i -o p -- $B > $out/p$i.log 2>&1
done
