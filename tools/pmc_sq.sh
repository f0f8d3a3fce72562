#!/bin/bash
# tools/pmc_sq.sh TAG CMD... -- the wavefront-level counters of a command in three rocprofv3 passes (no traffic counters):
# issue / wait / occupancy of every kernel.  Output: gpurun_out/TAG_sq.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
         "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcs_${TAG}_$i -o pmc -- "$@" > /dev/null 2>> $R/gpurun_out/pmcs_$TAG.log
  echo "pmc pass $i rc=$?"
done
python $R/tools/rocprof_summary.py pmc /tmp/pmcs_${TAG}_*/pmc_results.db 2>&1 | grep -E "^##|knn_|radius_|counter" > $R/gpurun_out/${TAG}_sq.txt
