#!/bin/bash
# Kernel ms of the k > 1 search by cap, batch size and k (BASELINE config 3 cloud, the first nq queries).
cd /root/repo
CAPS="PTK_KNN_CAP=8;PTK_KNN_CAP=12;PTK_KNN_CAP=16;PTK_KNN_CAP=24;PTK_KNN_CAP=32;PTK_KNN_CAP=48;PTK_KNN_CAP=64;PTK_KNN_CAP=96;PTK_KNN_CAP=128;PTK_KNN_CAP=192;PTK_KNN_CAP=256;PTK_KNN_CAP=384"
for k in ${KS:-16 8 4 32}; do for nq in ${NQS:-150000 300000 600000 900000 1200000 1800000 2400000 3600000}; do
  timeout 300 python tools/ab_env.py --configs ";$CAPS" --rounds 4 --k $k --nq $nq 2>&1 | tail -1
done; done
