#!/usr/bin/env python3
"""Host tree build + upload time of BASELINE config 2 for several PTK_BUILD_THREADS (the tree is identical)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

pts, _ = ds.config2_clouds("L", ds.CONFIG2_N, 1000)
ref = None
for th in (1, 8, 32, 32, 64, 128, 256):
    os.environ["PTK_BUILD_THREADS"] = str(th)
    t0 = time.perf_counter()
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    dt = time.perf_counter() - t0
    nodes, idx, _, _ = tree.flat()
    same = True if ref is None else bool(np.array_equal(nodes, ref[0]) and np.array_equal(idx, ref[1]))
    if ref is None:
        ref = (nodes, idx)
    print(f"threads {th:3d}: create (host build + encode + upload) {dt:.3f} s, identical to 1 thread: {same}", flush=True)
    tree.close()
