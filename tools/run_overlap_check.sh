#!/bin/bash
# The two capped launches side by side: A/B against one launch at several batch sizes, parity, fuzz.
cd /root/repo
for k in 16 8 4 32 2; do for nq in 1200000 2400000 4800000 7200863; do timeout 300 python tools/ab_env.py --configs ";PTK_KNN_OVERLAP_PCT=0;PTK_KNN_OVERLAP_PCT=0,PTK_KNN_CAP=256" --rounds 5 --k $k --nq $nq 2>&1 | tail -1; done; done > gpurun_out/ab_overlap_auto.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or coop or capped or config or sha or full or multi or replica" 2>&1 | grep -E "passed|failed"
PTK_KNN_CAP_MIN_NQ=1 timeout 600 python tools/fuzz_parity.py --cases 400 --seed 621 2>&1 | tail -1
