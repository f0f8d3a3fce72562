#!/bin/bash
# tools/profile_all.sh TAG -- rocprofv3 kernel-trace stats of the secondary benchmarks (config 3 and the forest).
# The headline benchmark has its own script (tools/pmc_run.sh).  Output: gpurun_out/TAG_*.txt / .json
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_c3 -o trace -- python $R/tools/bench_config3.py L > $R/gpurun_out/${TAG}_config3_L.jsonl 2> $R/gpurun_out/prof_${TAG}_c3.log
echo "config3 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_forest -o trace -- python $R/tools/bench_forest.py --cpu-queries 2000 > $R/gpurun_out/${TAG}_forest_bench.json 2> $R/gpurun_out/prof_${TAG}_forest.log
echo "forest rc=$?"
python $R/tools/rocprof_summary.py stats $R/gpurun_out/prof_${TAG}_c3/trace_results.db > $R/gpurun_out/${TAG}_config3_stats.txt 2>&1
python $R/tools/rocprof_summary.py stats $R/gpurun_out/prof_${TAG}_forest/trace_results.db > $R/gpurun_out/${TAG}_forest_stats.txt 2>&1
rm -rf $R/gpurun_out/prof_${TAG}_c3 $R/gpurun_out/prof_${TAG}_forest
