import sys, time, numpy as np, os
sys.path.insert(0, "/root/repo")
import torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
pts, q = ds.config2_clouds("L")
for grid in (0.25, 1.0):
    p2 = np.ascontiguousarray(np.round(pts / grid) * grid, dtype=np.float32)
    q2 = np.ascontiguousarray(np.round(q / grid) * grid, dtype=np.float32)
    tree = pt.KdTree(p2, pt.Metric.L2Squared, 10, device=0)
    dq = torch.from_numpy(q2).cuda()
    out = torch.empty((len(q2), 1, 2), dtype=torch.int32, device="cuda")
    for _ in range(2): tree.search_knn(dq, 1, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tree.search_knn(dq, 1, out)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"grid {grid}: {ms:.3f} ms per step, counts {tree.knn1_counts()}", flush=True)
