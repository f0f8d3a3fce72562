#!/bin/bash
# tools/prof_knn.sh TAG K [env...] -- rocprofv3 kernel-trace stats of a k-NN A/B line (tools/ab_env.py, one config).
TAG=$1; K=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o trace -- python $R/tools/ab_env.py --configs "X=1" --k $K --rounds 5 > $R/gpurun_out/${TAG}_ab.log 2> $R/gpurun_out/prof_${TAG}.log
python $R/tools/rocprof_summary.py stats $R/gpurun_out/prof_${TAG}/trace_results.db > $R/gpurun_out/${TAG}_stats.txt 2>&1
rm -rf $R/gpurun_out/prof_${TAG}
grep -E "knn_|kernel  " $R/gpurun_out/${TAG}_stats.txt | cut -c1-150
