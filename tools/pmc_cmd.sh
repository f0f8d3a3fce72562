#!/bin/bash
# tools/pmc_cmd.sh TAG CMD... -- rocprofv3 kernel-trace stats + PMC passes of an arbitrary command (one counter set per
# run, each under `timeout`).  Output: gpurun_out/TAG_stats.txt, TAG_pmc.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- "$@" > /dev/null 2> $R/gpurun_out/prof_$TAG.log
echo "trace rc=$?"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
         "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" \
         "SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${TAG}_$i -o pmc -- "$@" > /dev/null 2>> $R/gpurun_out/pmc_$TAG.log
  echo "pmc pass $i ($c) rc=$?"
done
python $R/tools/rocprof_summary.py stats $R/gpurun_out/prof_$TAG/trace_results.db > $R/gpurun_out/${TAG}_stats.txt 2>&1
python $R/tools/rocprof_summary.py pmc $R/gpurun_out/pmc_${TAG}_*/pmc_results.db > $R/gpurun_out/${TAG}_pmc.txt 2>&1
python $R/tools/rocprof_summary.py traffic $R/gpurun_out/pmc_${TAG}_1/pmc_results.db $R/gpurun_out/pmc_${TAG}_2/pmc_results.db > $R/gpurun_out/${TAG}_traffic.json 2>&1
rm -rf $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_${TAG}_*
