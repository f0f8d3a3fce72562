#!/usr/bin/env python3
"""ms per search by max_leaf_size (the reference's benchmarks use 10; a user may not): k = 1 / 16 / radius, device buffers."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

def ms(f):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 2)

pts, q = ds.config2_clouds("L", 3_000_000, 2_000_000)
dq = torch.from_numpy(q).cuda()
for leaf in (1, 2, 4, 10, 16, 32, 33, 64, 128):
    t0 = time.perf_counter()
    tree = pt.KdTree(pts, pt.Metric.L2Squared, leaf, device=0)
    create = round(time.perf_counter() - t0, 2)
    row = {"create_s": create, "depth": tree.info()["max_depth"]}
    for k in (1, 16):
        out = torch.empty((len(q), k, 2), dtype=torch.int32, device="cuda")
        row[f"knn{k}"] = ms(lambda: tree.search_knn(dq, k, out))
    row["radius0.25"] = ms(lambda: tree.search_radius_device(dq, 0.25))
    print("leaf", leaf, row, flush=True)
    tree.close()
