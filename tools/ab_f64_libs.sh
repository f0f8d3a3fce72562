#!/bin/bash
# tools/ab_f64_libs.sh LIB... -- builds of libptk.so ("-" = the tree's own) on one box: the double k-NN step over batch
# sizes (tools/time_f64_sizes.py), two passes, alternating.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pass in 1 2; do for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset PTK_LIBRARY; else export PTK_LIBRARY=$PWD/$lib; fi
  echo "== $lib pass $pass"; timeout 600 python tools/time_f64_sizes.py 2>&1 | grep float64
done; done
