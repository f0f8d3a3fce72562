mkdir -p gpurun_out
echo "== persistent variants, cloud L =="
python tools/ab_knn1.py --variants 1,2,4,8,10,12 --rounds 4 2>/dev/null | head -6
echo "== chunk sweep (variant 10) =="
for c in 256 4096; do echo "chunk=$c"; PTK_CHUNK=$c python tools/ab_knn1.py --variants 10 --rounds 3 2>/dev/null | head -1; done
echo "== cloud U =="
python tools/ab_knn1.py --variants 8,10,12 --rounds 3 --cloud U 2>/dev/null | head -3
