import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
pts, _ = ds.config2_clouds("L", ds.CONFIG2_N, 1000)
os.environ["PTK_CREATE_TIMING"] = os.environ.get("PTK_CREATE_TIMING", "1")
dev = 0 if pt.device_count() > 0 else pt.PTK_DEVICE_NONE
for i in range(2):
    t0 = time.perf_counter(); tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=dev); print("create", time.perf_counter() - t0, flush=True); del tree
