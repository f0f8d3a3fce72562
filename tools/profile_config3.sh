#!/bin/bash
# tools/profile_config3.sh TAG [env assignments...] -- rocprofv3 kernel-trace stats of tools/bench_config3.py on cloud L.
# Output: gpurun_out/TAG_config3_stats.txt and gpurun_out/TAG_config3_L.jsonl
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_c3 -o trace -- python $R/tools/bench_config3.py L > $R/gpurun_out/${TAG}_config3_L.jsonl 2> $R/gpurun_out/prof_${TAG}_c3.log
echo "config3 rc=$?"
python $R/tools/rocprof_summary.py stats $R/gpurun_out/prof_${TAG}_c3/trace_results.db > $R/gpurun_out/${TAG}_config3_stats.txt 2>&1
rm -rf $R/gpurun_out/prof_${TAG}_c3
