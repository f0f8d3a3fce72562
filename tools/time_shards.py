#!/usr/bin/env python3
"""Every shard of BASELINE configs[3] on one GPU, one after the other: the 7.2 M-query batch cut into 8 contiguous row
ranges (what ptk_multi_* / sharded.py give each GPU) and into 8 strided sets (rows r, r + 8, ...), whole tree, knn = 1.
ms per step of each: the slowest one is what an 8-GPU step takes before the gather."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
pts, q = ds.config2_clouds(sys.argv[1] if len(sys.argv) > 1 else "L")
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
parts = 8
per = (len(q) + parts - 1) // parts
for name, pick in (("contiguous", lambda r: q[r * per:(r + 1) * per]), ("strided", lambda r: q[r::parts])):
    times = []
    for r in range(parts):
        dq = torch.from_numpy(np.ascontiguousarray(pick(r))).cuda()
        out = torch.empty((len(dq), 1, 2), dtype=torch.int32, device="cuda")
        for _ in range(3): tree.search_knn(dq, 1, out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): tree.search_knn(dq, 1, out)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 30 * 1e3)
    print(f"{name}: " + " ".join(f"{t:.3f}" for t in times) + f"  ms; slowest {max(times):.3f}, mean {sum(times) / parts:.3f}", flush=True)
dq = torch.from_numpy(q).cuda(); out = torch.empty((len(q), 1, 2), dtype=torch.int32, device="cuda")
for _ in range(3): tree.search_knn(dq, 1, out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): tree.search_knn(dq, 1, out)
torch.cuda.synchronize(); print(f"whole batch: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
