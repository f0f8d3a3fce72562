import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
pts, q = ds.config2_clouds("L")
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
for k in (1, 16):
    out = np.empty((len(q), k) if k > 1 else (len(q),), dtype=pt.NEIGHBOR)
    tree.search_knn(q, k, out); tree.search_knn(q, k, out)
    os.environ["PTK_TEST_KNOBS"] = "host_trace=1"
    t0 = time.perf_counter(); tree.search_knn(q, k, out); print("k", k, (time.perf_counter() - t0) * 1e3, "ms", flush=True)
    os.environ["PTK_TEST_KNOBS"] = "host_trace=0"
