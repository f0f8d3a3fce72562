mkdir -p gpurun_out
echo "== E1 occupancy sweep (variant 8: S=32 B=64, 16KB/block) =="
for x in 0 16384 49152 114688; do echo "extra_lds=$x"; PTK_EXTRA_LDS=$x python tools/ab_knn1.py --variants 8 --rounds 3 2>/dev/null | head -1; done
echo "== E2 nq scaling (variant 8) =="
for nq in 900000 1800000 3600000 7200863; do echo "nq=$nq"; python tools/ab_knn1.py --variants 8 --rounds 3 --nq $nq 2>/dev/null | head -1; done
echo "== E3 order: morton-presorted input =="
python tools/ab_knn1.py --variants 8 --rounds 3 --order morton 2>/dev/null | head -1
echo "== E4 cloud U =="
python tools/ab_knn1.py --variants 1,8 --rounds 3 --cloud U 2>/dev/null | head -2
