#!/usr/bin/env python3
"""Where the host-buffer box search spends its time (2 M boxes, cloud L): the binding's strided slicing, the C call
(kernels + allocations + pageable copies) and freeing the 274 MB result.  Measured: 13.6 / 35.1 / 17.6 ms."""
import sys, time, ctypes, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
from ctypes import c_void_p, byref
pts, q = ds.config2_clouds("L"); q = q[:2_000_000]
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
boxes = np.empty((2 * len(q), 3), dtype=np.float32); boxes[0::2], boxes[1::2] = q - np.float32(.5), q + np.float32(.5)
tree.search_box(boxes)
t0 = time.perf_counter(); mins = np.ascontiguousarray(boxes[0::2]); maxs = np.ascontiguousarray(boxes[1::2]); t1 = time.perf_counter()
lib = pt._load(); off = np.zeros(len(q) + 1, dtype=np.uint64); rows = c_void_p()
lib.ptk_search_box(tree._h, mins.ctypes.data, maxs.ctypes.data, len(q), off.ctypes.data, byref(rows)); lib.ptk_free(rows)
t2 = time.perf_counter(); rows = c_void_p()
lib.ptk_search_box(tree._h, mins.ctypes.data, maxs.ctypes.data, len(q), off.ctypes.data, byref(rows)); t3 = time.perf_counter()
lib.ptk_free(rows); t4 = time.perf_counter()
print("slicing ms", (t1 - t0) * 1e3, "C call ms", (t3 - t2) * 1e3, "free ms", (t4 - t3) * 1e3, "rows MB", off[-1] * 4 / 1e6)
