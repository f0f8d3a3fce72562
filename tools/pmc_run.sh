#!/bin/bash
# tools/pmc_run.sh TAG [bench args...] -- rocprofv3 kernel-trace stats + PMC passes of bench.py.
# Each --pmc set is collected in its own run with --kernel-trace only (gpurun refuses other trace domains
# together with counters).  Every pass runs under `timeout`: some TA/TCP counter sets stall this pool.
# Output: gpurun_out/{prof,pmc}_TAG*/ ; summarise with tools/rocprof_summary.py.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pipelined "$@" > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.log
echo "trace rc=$?"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
         "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${TAG}_$i -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined "$@" > /dev/null 2>> $R/gpurun_out/pmc_$TAG.log
  echo "pmc pass $i ($c) rc=$?"
done
# Summarise on the box and drop the (large) databases: gpurun copies back at most 64 MiB.
python $R/tools/rocprof_summary.py stats $R/gpurun_out/prof_$TAG/trace_results.db > $R/gpurun_out/${TAG}_stats.txt 2>&1
python $R/tools/rocprof_summary.py pmc $R/gpurun_out/pmc_${TAG}_*/pmc_results.db > $R/gpurun_out/${TAG}_pmc.txt 2>&1
python $R/tools/rocprof_summary.py traffic $R/gpurun_out/pmc_${TAG}_1/pmc_results.db $R/gpurun_out/pmc_${TAG}_2/pmc_results.db > $R/gpurun_out/${TAG}_traffic.json 2>&1
python $R/tools/rocprof_summary.py timeline $R/gpurun_out/prof_$TAG/trace_results.db > $R/gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_${TAG}_*
