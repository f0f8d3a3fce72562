#!/usr/bin/env python3
"""Radius search (count + scan + fill on device buffers) of the first nq queries of BASELINE config 2/3, ms per step.
python tools/time_radius.py [nq ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
pts, q = ds.config2_clouds("L")
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
for nq in [int(a) for a in sys.argv[1:]] or [len(q)]:
    dq = torch.from_numpy(np.ascontiguousarray(q[:nq])).cuda()
    off, raw = tree.search_radius_device(dq, 1.0)
    torch.cuda.synchronize()
    tree.profile(enable=True, reset=True)
    t0 = time.perf_counter()
    for _ in range(3):
        off, raw = tree.search_radius_device(dq, 1.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    p = tree.profile(enable=False, reset=True)
    print(f"nq {nq}: {ms:.3f} ms per step, kernels {p['search_ms'] / 3:.3f} ms, hits {int(off[-1])}", flush=True)
