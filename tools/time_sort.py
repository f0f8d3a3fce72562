#!/usr/bin/env python3
"""The batch order of a k = 1 search against the batch size, for the three sorts that can make it (test hook sort = 0: rocprim's
onesweep; 1: the library's passes with one wavefront per tile; sort_block = 1: with blocks of eight wavefronts on tiles of 4 096 rows):
reorder ms and step ms per size (HIP events inside the library).  python tools/time_sort.py [sizes ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

pts, q = ds.config2_clouds("L")
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
sizes = [int(a) for a in sys.argv[1:]] or [100_000, 300_000, 900_108, 2_000_000, 3_600_000, len(q)]
forms = {"rocprim": {"PTK_TEST_KNOBS": "sort=0,sort_block=0"}, "one wavefront per tile": {"PTK_TEST_KNOBS": "sort=1,sort_block=0"},
         "blocks of eight wavefronts": {"PTK_TEST_KNOBS": "sort=1,sort_block=1"}}
for nq in sizes:
    dq = torch.from_numpy(np.ascontiguousarray(q[:nq])).cuda()
    out = torch.empty((nq, 1, 2), dtype=torch.int32, device="cuda")
    line = []
    for name, env in forms.items():
        os.environ.update(env)
        for _ in range(3):
            tree.search_knn(dq, 1, out)
        torch.cuda.synchronize()
        tree.profile(enable=True, reset=True)
        t0 = time.perf_counter()
        for _ in range(20):
            tree.search_knn(dq, 1, out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        p = tree.profile(enable=False, reset=True)
        line.append(f"{name}: reorder {p['reorder_ms'] / 20:.4f} step {ms:.4f}")
    print(f"nq {nq:8d}  " + " | ".join(line), flush=True)
