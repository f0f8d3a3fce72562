#!/bin/bash
# tools/trace_run.sh TAG [bench args...] -- rocprofv3 --kernel-trace --stats of a short bench.py run (no counters).
# Environment knobs of libptk (PTK_*) pass through.  Output: gpurun_out/TAG_stats.txt, TAG_timeline.txt, TAG_bench.json
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pipelined "$@" > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.log
echo "trace rc=$?"
python $R/tools/rocprof_summary.py stats $R/gpurun_out/prof_$TAG/trace_results.db > $R/gpurun_out/${TAG}_stats.txt 2>&1
python $R/tools/rocprof_summary.py timeline $R/gpurun_out/prof_$TAG/trace_results.db > $R/gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf $R/gpurun_out/prof_$TAG
