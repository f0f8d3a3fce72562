#!/bin/bash
# tools/pmc_forest.sh TAG -- counters of the kd-forest search of BASELINE configs[4] (tools/bench_forest.py): FETCH_SIZE,
# WRITE_SIZE, L2 hits / misses and the issue counters of forest_knn_kernel, each in a rocprofv3 pass of its own.
# Output: gpurun_out/TAG_forest_pmc.txt
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcf_${TAG}_$i -o pmc -- python $R/tools/bench_forest.py --cpu-queries 100 > /dev/null 2>> $R/gpurun_out/pmcf_$TAG.log
  echo "pmc pass $i ($c) rc=$?"
done
python $R/tools/rocprof_summary.py pmc /tmp/pmcf_${TAG}_*/pmc_results.db 2>&1 | grep -E "^##|forest_|counter" > $R/gpurun_out/${TAG}_forest_pmc.txt
