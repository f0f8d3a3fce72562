import sys, os, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch, pico_tree_amd as pt
from pico_tree_amd import datasets as ds
pts, q = ds.config2_clouds("L")
for grid in (0.25, 1.0):
    p2 = np.ascontiguousarray(np.round(pts / grid) * grid, dtype=np.float32)
    t0 = time.time(); tree = pt.KdTree(p2, pt.Metric.L2Squared, 10, device=0); print("create", round(time.time() - t0, 3), tree.piles(), tree.info()["max_depth"], flush=True)
