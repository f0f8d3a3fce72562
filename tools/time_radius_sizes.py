#!/usr/bin/env python3
"""Kernel ms of the radius search (count pass | fill pass) on BASELINE config 3's cloud against the batch size and the
cap of the list pass (test hook radius_cap; 0 = every query to its end in its lane, -1 = the rule of the batch).

    python tools/time_radius_sizes.py [caps, e.g. 0,-1,16,32,64] [sizes, e.g. 20000,150000,900108,7200863]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

caps = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "0,-1").split(",")]
sizes = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "20000,150000,900108,7200863").split(",")]
pts, q = ds.config2_clouds("L", ds.CONFIG2_N, ds.CONFIG2_NQ)
tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
dq = torch.from_numpy(q).cuda()
lib = pt._load()
stream = torch.cuda.current_stream().cuda_stream
for nq in sizes:
    d = dq[:nq]
    counts = torch.zeros(nq + 1, dtype=torch.int64, device="cuda")
    base = None
    for cap in caps:
        pt.set_test_knobs(radius_cap=None if cap < 0 else cap)
        best = [1e9, 1e9]
        for rep in range(4):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            pt._check(lib.ptk_search_radius_count_device(tree._h, d.data_ptr(), nq, np.float32(1.0), np.float32(1.0), counts.data_ptr(), stream))
            ev[1].record()
            offsets = torch.zeros(nq + 1, dtype=torch.int64, device="cuda")
            offsets[1:] = torch.cumsum(counts[:nq], 0)
            total = int(offsets[-1].item())
            out = torch.empty((max(total, 1), 2), dtype=torch.int32, device="cuda")
            ev2 = torch.cuda.Event(enable_timing=True); ev2.record()
            pt._check(lib.ptk_search_radius_fill_device(tree._h, d.data_ptr(), nq, np.float32(1.0), np.float32(1.0), offsets.data_ptr(), out.data_ptr(), 0, stream))
            ev[2].record()
            torch.cuda.synchronize()
            best[0] = min(best[0], ev[0].elapsed_time(ev[1])); best[1] = min(best[1], ev2.elapsed_time(ev[2]))
        h = hash(out.cpu().numpy().tobytes())
        if base is None: base = h
        c = tree.radius_coop_counts()
        print(f"nq {nq:8d} cap {cap:4d}: count {best[0]:7.3f} ms  fill {best[1]:7.3f} ms  sum {best[0]+best[1]:7.3f}  hits {total}  coop {c}  rows {'same' if h == base else 'DIFFER'}", flush=True)
        del out, offsets
pt.set_test_knobs()
