mkdir -p gpurun_out
echo "== two-phase variants, cloud L =="
python tools/ab_knn1.py --variants 4,21,23,30,31,32 --rounds 4 2>/dev/null | head -5
echo "== cloud U =="
python tools/ab_knn1.py --variants 4,21,23,30,31,32 --rounds 3 --cloud U 2>/dev/null | head -5
