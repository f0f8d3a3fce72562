#!/bin/bash
# tools/tcc_ab.sh LIB... -- L2 hit / miss counters (one rocprofv3 --pmc pass) of the k = 1 search of cloud L with each of
# exp_libs/LIB swapped in ("main" = the library as built).  Output: gpurun_out/tcc_ab_LIB.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" = main ]; then RUN=""; else RUN="bash $R/tools/exp_lib.sh $L"; fi
  $RUN timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/tcc_$L -o pmc -- python $R/tools/ab_env.py --configs "PTK_X=0" --rounds 3 --cloud L --k 1 > /dev/null 2>&1
  echo "== $L"
  python $R/tools/rocprof_summary.py pmc /tmp/tcc_$L/pmc_results.db 2>&1 | grep "knn1_phase\|knn1_coop" | cut -c1-150 | tee $R/gpurun_out/tcc_ab_$L.txt
done
