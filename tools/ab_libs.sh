#!/bin/bash
# A/B of whole-library variants (tools/bin/libptk_<name>.so, built by hand with extra -D flags): one process each,
# the same step of BASELINE config 3 (k from $KS), two passes so that drift between processes shows.
#   tools/ab_libs.sh "default eu5 eu6 eu8"
cd /root/repo
for pass in 1 2; do
  for name in $1; do
    for k in ${KS:-16 4}; do
      if [ "$name" = default ]; then unset PTK_LIBRARY; else export PTK_LIBRARY=/root/repo/tools/bin/libptk_$name.so; fi
      echo "== $name k=$k pass $pass"
      timeout 300 python tools/ab_env.py --configs "PTK_KNN_CAP=256" --rounds 5 --k $k 2>&1 | tail -1
    done
  done
done
