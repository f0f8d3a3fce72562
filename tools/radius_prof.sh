#!/bin/bash
# tools/radius_prof.sh TAG -- the radius search of BASELINE config 3 (tools/time_radius.py): kernel durations
# (rocprofv3 --kernel-trace --stats) and FETCH_SIZE / WRITE_SIZE / issue counters in passes of their own.
# Output: gpurun_out/TAG_radius_stats.txt, gpurun_out/TAG_radius_pmcq.txt
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace -- python $R/tools/time_radius.py > $R/gpurun_out/${TAG}_radius.log 2>/dev/null
python $R/tools/rocprof_summary.py stats /tmp/prof_$TAG/trace_results.db 2>&1 | grep -E "radius|kernel" | head -8 > $R/gpurun_out/${TAG}_radius_stats.txt
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcr_${TAG}_$i -o pmc -- python $R/tools/time_radius.py > /dev/null 2>&1
done
python $R/tools/rocprof_summary.py pmc /tmp/pmcr_${TAG}_*/pmc_results.db 2>&1 | grep -E "^##|radius_|counter" > $R/gpurun_out/${TAG}_radius_pmcq.txt
cat $R/gpurun_out/${TAG}_radius.log $R/gpurun_out/${TAG}_radius_stats.txt $R/gpurun_out/${TAG}_radius_pmcq.txt
