#!/bin/bash
# The auto cap of the k > 1 searches on the MI355X: parity (suite + fuzz with the cap on every batch), A/B, bench.
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or coop or capped or config or sha or full" 2>&1 | tail -3
PTK_KNN_CAP_MIN_NQ=1 timeout 600 python tools/fuzz_parity.py --cases 500 --seed 611 2>&1 | tail -3
PTK_KNN_CAP_MIN_NQ=1 PTK_KNN_COOP_WAVES=1 timeout 600 python tools/fuzz_parity.py --cases 300 --seed 612 2>&1 | tail -3
for k in 8 32; do for nq in 20000 150000 600000 2400000; do timeout 300 python tools/ab_env.py --configs ";PTK_KNN_CAP=256" --rounds 5 --k $k --nq $nq 2>&1 | tail -1; done; done > gpurun_out/ab_cap_auto2.jsonl
for k in 16 8 4; do echo "k=$k"; timeout 400 python tools/ab_host.py --k $k --rounds 4 --configs ";PTK_HOST_PIECE=2400288" 2>&1 | tail -2; done
