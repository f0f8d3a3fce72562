#!/bin/bash
# tools/shard_prof.sh TAG [env...] -- rocprofv3 kernel trace of tools/shard_step.py: gpurun_out/TAG_shard_stats.txt, TAG_shard_timeline.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- python $R/tools/shard_step.py --rounds 10 > $R/gpurun_out/${TAG}_shard.log 2> $R/gpurun_out/prof_$TAG.log
python $R/tools/rocprof_summary.py stats $R/gpurun_out/prof_$TAG/trace_results.db > $R/gpurun_out/${TAG}_shard_stats.txt 2>&1
python $R/tools/rocprof_summary.py timeline $R/gpurun_out/prof_$TAG/trace_results.db > $R/gpurun_out/${TAG}_shard_timeline.txt 2>&1
rm -rf $R/gpurun_out/prof_$TAG
cat $R/gpurun_out/${TAG}_shard_timeline.txt
