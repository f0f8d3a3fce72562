import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
pts, q = ds.config2_clouds("L")
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
nq = 225027
dq = torch.from_numpy(np.ascontiguousarray(q[:nq])).cuda()
off, raw = tree.search_radius_device(dq, 1.0)
c = np.diff(off.cpu().numpy())
print("hits: mean", c.mean(), "p50", np.percentile(c, 50), "p99", np.percentile(c, 99), "p99.9", np.percentile(c, 99.9), "max", c.max())
for _ in range(3):
    off, raw = tree.search_radius_device(dq, 1.0)
torch.cuda.synchronize()
