#!/bin/bash
# tools/exp_lib.sh LIB CMD... -- run CMD with exp_libs/LIB swapped in for libptk.so (on the GPU box's copy).
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$1; shift
cp $R/pico_tree_amd/csrc/libptk.so /tmp/libptk_keep.so
cp $R/exp_libs/$L $R/pico_tree_amd/csrc/libptk.so
"$@"
cp /tmp/libptk_keep.so $R/pico_tree_amd/csrc/libptk.so
