"""Step time of the double radius search (host entry: both passes + the copies) over caps (test hook radius64_cap; 0 =
uncapped) and batch sizes, BASELINE config 2's cloud L in float64, r = 1."""
import sys, os, time, json, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
p, q = ds.config2_clouds("L")
tree = pt.KdTree(p.astype(np.float64), pt.Metric.L2Squared, 10, device=0)
for nq in (2000, 20000, 150000, 600000):
    qq = np.ascontiguousarray(q[:: len(q) // nq][:nq].astype(np.float64))
    row = {"nq": len(qq)}
    for cap in (0, 4, 8, 16, 32, 64, 128, 256):
        pt.set_test_knobs(radius64_cap=cap)
        tree.search_radius(qq, 1.0)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); tree.search_radius(qq, 1.0); ts.append((time.perf_counter() - t0) * 1e3)
        c = tree.knn_coop_counts()
        row[str(cap)] = [round(sorted(ts)[1], 2), c["cooperative"], c["redone"]]
    pt.set_test_knobs(radius64_cap=None)
    print(json.dumps(row), flush=True)
