#!/bin/bash
# tools/pmc_quick.sh TAG CMD... -- three counter passes (known-good counters only: a pass with TCC_EA0_* counters hung a box for 15 minutes)
# three counter passes (FETCH_SIZE | WRITE_SIZE | wait / issue counters) of a command.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcq_${TAG}_$i -o pmc -- "$@" > /dev/null 2>> $R/gpurun_out/pmcq_$TAG.log
  echo "pmc pass $i ($c) rc=$?"
done
python $R/tools/rocprof_summary.py pmc /tmp/pmcq_${TAG}_*/pmc_results.db > $R/gpurun_out/${TAG}_pmcq.txt 2>&1
