cd /root/repo
for k in 40 64; do for nq in 150000 900000 7200863; do timeout 300 python tools/ab_env.py --configs ";PTK_KNN_CAP=0;PTK_KNN_CAP=256;PTK_KNN_CAP=512" --rounds 3 --k $k --nq $nq 2>&1 | tail -1; done; done > gpurun_out/ab_k64.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knn or coop or capped or config or sha or full or cap_follows" 2>&1 | grep -E "passed|failed"
timeout 900 python tools/fuzz_parity.py --cases 600 --seed 711 2>&1 | tail -2
PTK_KNN_CAP_MIN_NQ=1 timeout 900 python tools/fuzz_parity.py --cases 500 --seed 712 2>&1 | tail -2
