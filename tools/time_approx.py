#!/usr/bin/env python3
"""Approximate k = 1 searches (search_knn(pts, 1, e)) of BASELINE config 2 on the device: ms per step for several e.
An approximate search has no cap and no cooperative tail (its answer depends on the visit order: DESIGN.md section 9)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
cloud = sys.argv[1] if len(sys.argv) > 1 else "L"
pts, q = ds.config2_clouds(cloud)
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
dq = torch.from_numpy(q).cuda()
out = torch.empty((len(q), 1, 2), dtype=torch.int32, device="cuda")
for e in (1.0, 1.05, 1.5, 2.0):
    args = (dq, 1, out) if e == 1.0 else (dq, 1, e, out)
    tree.search_knn(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tree.search_knn(*args)
    torch.cuda.synchronize()
    print(f"cloud {cloud} e = {e}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per step", flush=True)
