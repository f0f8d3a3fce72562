import sys, os, time, json, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
p, q = ds.config2_clouds("L")
def med(fn, reps=9):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(sorted(ts)[len(ts) // 2], 3)
for dtype in (np.float64, np.float32):
    tree = pt.KdTree(p.astype(dtype), pt.Metric.L2Squared, 10, device=0)
    for nq in (8, 64, 200):
        for off in (0, 3):
            qq = np.ascontiguousarray(q[off:: len(q) // nq][:nq].astype(dtype))
            row = {"dtype": np.dtype(dtype).name, "nq": nq, "off": off}
            for name, knobs in (("rule", {}), ("min1", {"knn_cap_min_nq": 1})):
                pt.set_test_knobs(**knobs)
                row[name] = [med(lambda: tree.search_knn(qq, 1)), med(lambda: tree.search_knn(qq, 16)), med(lambda: tree.search_radius(qq, 1.0))]
                pt.set_test_knobs(**{k: None for k in knobs})
            print(json.dumps(row), flush=True)
