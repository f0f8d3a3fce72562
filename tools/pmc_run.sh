#!/bin/bash
# tools/pmc_run.sh TAG [bench args...] -- rocprofv3 kernel-trace stats + PMC passes of bench.py.
# Each --pmc set is collected in its own run with --kernel-trace only (gpurun refuses other trace domains
# together with counters).  Output: gpurun_out/{prof,pmc}_TAG*/ ; summarise with tools/rocprof_summary.py.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.log
echo "trace rc=$?"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
         "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${TAG}_$i -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>> $R/gpurun_out/pmc_$TAG.log
  echo "pmc pass $i ($c) rc=$?"
done
