#!/bin/bash
# tools/knn_ab.sh K LIB... -- tools/ab_env.py (cloud L and U, knn = K) with each of exp_libs/LIB swapped in ("main" = as built)
R=${GRAFT_REPO_ROOT:-/root/repo}
K=$1; shift
for L in "$@"; do
  if [ "$L" = main ]; then RUN=""; else RUN="bash $R/tools/exp_lib.sh $L"; fi
  echo "== $L"
  for C in L U; do
    $RUN python $R/tools/ab_env.py --configs "PTK_X=0" --rounds 5 --cloud $C --k $K 2>/dev/null | tail -1 | cut -c1-260
  done
done
