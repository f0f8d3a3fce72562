#!/usr/bin/env python3
"""How long do the most expensive queries of cloud L take on their own?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

pts, q = ds.config2_clouds("L")
ref = oracle.Oracle(pts, 10, "port")
_, cnt = ref.search_knn(q, 1, counters=True)
cost = cnt[:, 0].astype(np.int64) + 3 * cnt[:, 1].astype(np.int64)
order = np.argsort(-cost)
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
tree.set_reorder(pt.REORDER_OFF)

def run(sel, label, variant="4"):
    os.environ["PTK_KNN1_VARIANT"] = variant
    dq = torch.from_numpy(np.ascontiguousarray(q[sel])).cuda()
    out = torch.empty((len(sel), 1, 2), dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(4):
        tree.profile(enable=True, reset=True)
        tree.search_knn(dq, 1, out)
        torch.cuda.synchronize()
        ts.append(tree.profile(enable=False, reset=True)["search_ms"])
    print(f"{label:40s} n={len(sel):8d} variant {variant}: kernel ms {min(ts):.4f}  (max cost {cost[sel].max()}, mean {cost[sel].mean():.1f})", flush=True)

for v in ("0", "4"):
    run(order[:1], "the single worst query", v)
    run(order[:64], "worst 64 (one wave)", v)
    run(order[:1024], "worst 1024", v)
    run(order[100_000:100_064], "64 typical-ish (rank 100k)", v)
    run(order[-64:], "cheapest 64", v)
    cheap = np.sort(order[2000:])   # everything except the 2000 worst, original order
    tree.set_reorder(pt.REORDER_ON)
    run(cheap, "all but the worst 2000", v)
    run(np.arange(len(q)), "all", v)
    # monsters first: worst 4096 in front, then the rest in Morton order (reorder off)
    tree.set_reorder(pt.REORDER_OFF)
    rest = np.sort(order[4096:])
    rest = rest[ds.morton_order(q[rest])]
    run(np.concatenate([order[:4096], rest]), "worst 4096 first, rest Morton", v)
    tree.set_reorder(pt.REORDER_OFF)
