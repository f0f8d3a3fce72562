#!/usr/bin/env python3
"""Prints the instrumentation counters of the refill phase-2 kernel (PTK_DEBUG_STATS=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

cloud = sys.argv[1] if len(sys.argv) > 1 else "L"
pts, q = ds.config2_clouds(cloud, ds.CONFIG2_N, ds.CONFIG2_NQ)
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
dq = torch.from_numpy(q).cuda()
os.environ["PTK_DEBUG_STATS"] = "1"
for v in (sys.argv[2] if len(sys.argv) > 2 else "40").split(","):
    os.environ["PTK_KNN1_VARIANT"] = v
    tree.search_knn(dq, 1)
    torch.cuda.synchronize()
