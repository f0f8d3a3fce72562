cd /root/repo
for k in 16 8 4 2 32; do for nq in 20000 150000 600000 2400000 7200863; do timeout 300 python tools/ab_env.py --configs ";PTK_KNN_CAP=256" --rounds 5 --k $k --nq $nq 2>&1 | tail -1; done; done > gpurun_out/ab_cap_auto.jsonl
for k in 16 8 4; do echo "k=$k"; timeout 400 python tools/ab_host.py --k $k --rounds 4 --configs ";PTK_HOST_PIECE=600072;PTK_HOST_PIECE=600072,PTK_HOST_STREAMS=1;PTK_HOST_PIECE=900108;PTK_HOST_PIECE=1200144;PTK_HOST_PIECE=1200144,PTK_HOST_STREAMS=1;PTK_HOST_PIECE=2400288" 2>&1 | tail -7; done > gpurun_out/ab_host_piece2.txt 2>&1
