#!/bin/bash
# tools/radius_ab.sh LIB... -- per-kernel times (rocprofv3 --kernel-trace --stats) of tools/time_radius.py with each of
# exp_libs/LIB swapped in for libptk.so ("main" = the library as built).  Output: gpurun_out/radius_ab_LIB.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" = main ]; then RUN=""; else RUN="bash $R/tools/exp_lib.sh $L"; fi
  $RUN timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab_$L -o trace -- python $R/tools/time_radius.py > /tmp/ab_$L.out 2> /tmp/ab_$L.err
  echo "== $L: $(tail -1 /tmp/ab_$L.out)"
  python $R/tools/rocprof_summary.py stats /tmp/prof_ab_$L/trace_results.db 2>&1 | grep "radius\|kernel  " | cut -c1-150 | tee $R/gpurun_out/radius_ab_$L.txt
done
