#!/bin/bash
# tools/ab_libs.sh LIB_A LIB_B -- two builds of libptk.so on one box, one process per library and pass, alternating
# (tools/ab_env.py inside): the headline step (k = 1), knn = 16 and a shard of configs[3].  "-" = the tree's own library.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pass in 1 2 3; do for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset PTK_LIBRARY; else export PTK_LIBRARY=$PWD/$lib; fi
  for k in 1 16; do echo "== $lib k=$k pass $pass: $(timeout 300 python tools/ab_env.py --configs ";" --rounds 5 --k $k 2>&1 | tail -1 | cut -c1-200)"; done
  echo "== $lib shard pass $pass: $(timeout 300 python tools/ab_env.py --configs ";" --rounds 7 --k 1 --nq 900000 2>&1 | tail -1 | cut -c1-200)"
done; done
