#!/usr/bin/env python3
"""Experiment: BASELINE config 2 with the coordinates of points and queries snapped to a grid (exact ties between
distances, coincident points): step time, how many queries go through the cooperative search and the replay, parity
on a sample.  python tools/exp_quantised.py"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
import oracle
pts, q = ds.config2_clouds("L")
# (grid, shift): the queries are snapped to the same grid and then moved by shift x grid along every axis -- 0: most
# queries coincide with a pile of points (distance 0); 0.3: none does, every pile is at a distance > 0
for grid, shift in ((0.05, 0.0), (0.1, 0.0), (0.25, 0.0), (1.0, 0.0), (0.25, 0.3), (1.0, 0.3)):
    p2 = np.ascontiguousarray(np.round(pts / grid) * grid, dtype=np.float32)
    q2 = np.ascontiguousarray(np.round(q / grid) * grid + np.float32(shift * grid), dtype=np.float32)
    tree = pt.KdTree(p2, pt.Metric.L2Squared, 10, device=0)
    dq = torch.from_numpy(q2).cuda()
    out = torch.empty((len(q2), 1, 2), dtype=torch.int32, device="cuda")
    for _ in range(2): tree.search_knn(dq, 1, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tree.search_knn(dq, 1, out)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    res = pt.DeviceNeighbors(out).numpy()
    sample = np.linspace(0, len(q2) - 1, 20000, dtype=np.int64)
    ref = oracle.Oracle(p2, 10, "port")
    want = ref.search_knn(q2[sample], 1)
    ok = res[sample][:, None].tobytes() == want.tobytes()
    ref.close()
    print(f"grid {grid} shift {shift}: depth {tree.info()['max_depth']}, piles {tree.piles()}, {ms:.3f} ms per step, counts {tree.knn1_counts()}, parity on 20 k sample {ok}", flush=True)
    del tree
