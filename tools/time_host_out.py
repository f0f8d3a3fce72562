#!/usr/bin/env python3
"""Host-buffer k = 1 search of BASELINE config 2: result array allocated by every call vs handed in (search_knn(q, 1, nns))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
pts, q = ds.config2_clouds("L")
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
def run(f, n=6):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
fresh = run(lambda: tree.search_knn(q, 1))
out = np.empty((len(q), 1), dtype=pt.NEIGHBOR)
out[:] = 0
given = run(lambda: tree.search_knn(q, 1, out))
t0 = time.perf_counter(); a = np.empty((len(q), 1), dtype=pt.NEIGHBOR); a[:] = 0; touch = (time.perf_counter() - t0) * 1e3
print(f"result array per call {fresh:.2f} ms, handed in {given:.2f} ms; allocating + touching 58 MB on this host: {touch:.2f} ms")
