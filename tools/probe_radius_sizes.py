#!/usr/bin/env python3
"""ms per radius search by radius (config 3's cloud, 1 M queries): the rows grow from tens to thousands of hits; a
query whose list of leaves outgrows 1 024 entries is served by the traversal fill pass."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

pts, q = ds.config2_clouds("L", ds.CONFIG2_N, 1_000_000)
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
dq = torch.from_numpy(q).cuda()
for r2 in (0.01, 0.25, 1.0, 4.0, 16.0, 36.0):
    off, raw = tree.search_radius_device(dq, r2); torch.cuda.synchronize()
    hits = int(off[-1].item()); del off, raw
    t0 = time.perf_counter(); off, raw = tree.search_radius_device(dq, r2); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    mx = int((off[1:] - off[:-1]).max().item()); del off, raw
    print(f"r2 {r2}: {ms:.2f} ms, {hits / len(q):.1f} hits/query (max {mx}), {hits * 8 / ms / 1e6:.1f} GB/s of rows", flush=True)
