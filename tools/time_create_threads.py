#!/usr/bin/env python3
"""Creation time of the BASELINE config 2 tree for several PTK_BUILD_THREADS (device-assisted build), steady state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
pts, _ = ds.config2_clouds("L", ds.CONFIG2_N, 1000)
pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0).close()
for th in sys.argv[1:] or ["16", "32", "64", "128", "256"]:
    os.environ["PTK_BUILD_THREADS"] = th
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
        best = min(best, time.perf_counter() - t0)
        ph = tree.create_phases()
        tree.close()
    print(f"threads {th:>3}: create {best * 1e3:7.1f} ms (last: {ph})", flush=True)
