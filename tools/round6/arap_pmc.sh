#!/bin/bash
# round 6: a NAMED cause for config 4's roofline fraction -- SQ / TCC / TCP counter passes of the two kernels of an ARAP PCG iteration
# (arap_applySym, arap_flatStepRec).  Every pass in its own rocprofv3 run, counters only (no trace domains), wrapped in `timeout`.
#   bash tools/round6/arap_pmc.sh <outdir> [config]      (config: an OPT_AMD_CONFIG selector of tools/bench_configs.py, default config4)
out=$1; cfg=${2:-config4}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > $out/counters_available.txt 2>&1
have() { grep -qw "$1" $out/counters_available.txt; }
pick() { local s=""; for c in "$@"; do if have $c; then s="$s $c"; fi; done; echo $s; }
export OPT_AMD_CONFIG="$cfg" OPT_AMD_NO_TIMING_RUN=1
B="python tools/bench_configs.py"
i=0
for set in "$(pick SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU)" \
           "$(pick SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS)" \
           "$(pick TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum)" \
           "$(pick TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum)" \
           "$(pick TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum)" \
           "$(pick TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum)" \
           "$(pick TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum)" \
           "$(pick GRBM_GUI_ACTIVE GRBM_COUNT)"; do
  [ -z "$set" ] && continue
  i=$((i+1))
  echo "pass $i: $set" >> $out/passes.txt
  timeout 200 rocprofv3 --pmc $set -f csv -d $out/p$i -o p -- $B > $out/p$i.log 2>&1
done
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $out/kt -o kt -- $B > $out/kt.log 2>&1
unset OPT_AMD_CONFIG OPT_AMD_NO_TIMING_RUN
python tools/round6/pmc_by_kernel.py $out > $out/pmc_by_kernel.txt 2>&1
# the raw per-dispatch rows are tens of MiB per pass (gpurun brings back 64 MiB at most): keep the table, the pass list, the logs and the kernel-trace statistics
mkdir -p $out/kt_stats; find $out/kt -name "*stats*.csv" -exec cp {} $out/kt_stats/ \; 2>/dev/null
rm -rf $out/p[0-9]* $out/kt $out/counters_available.txt
cat $out/pmc_by_kernel.txt
