"""ms per HOST-buffer call of the double tree on small batches (numpy in, numpy out): knn = 1 / 16 and radius r = 1 on
BASELINE config 2's cloud L."""
import sys, os, time, json, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
p, q = ds.config2_clouds("L")
tree = pt.KdTree(p.astype(np.float64), pt.Metric.L2Squared, 10, device=0)
def med(fn, reps=9):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(sorted(ts)[len(ts) // 2], 3)
for nq in (64, 2000, 20000):
    qq = np.ascontiguousarray(q[:: len(q) // nq][:nq].astype(np.float64))
    print(json.dumps({"nq": nq, "knn1_ms": med(lambda: tree.search_knn(qq, 1)), "knn16_ms": med(lambda: tree.search_knn(qq, 16)),
                      "radius_ms": med(lambda: tree.search_radius(qq, 1.0))}), flush=True)
