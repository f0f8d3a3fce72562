#!/usr/bin/env python3
"""CPU-only analysis: per-query traversal cost (reference visit counters) against what phase 1
knows about the query (continuation class, home-leaf best distance).  Answers: can the expensive
queries be recognised before phase 2?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import oracle
from pico_tree_amd import datasets as ds
from tests.emu import EmulatedTree

cloud = sys.argv[1] if len(sys.argv) > 1 else "L"
nsample = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
pts, q = ds.config2_clouds(cloud, ds.CONFIG2_N, ds.CONFIG2_NQ)
rng = np.random.default_rng(3)
sel = np.sort(rng.choice(len(q), nsample, replace=False))
qs = np.ascontiguousarray(q[sel])
ref = oracle.Oracle(pts, 10, "port")
ref.set_threads(8)
_, cnt = ref.search_knn(qs, 1, counters=True)
cost = cnt[:, 0].astype(np.int64) + 2 * cnt[:, 1].astype(np.int64)   # node steps + ~2 batches per leaf
emu = EmulatedTree(pts, 10)
cls = np.zeros(nsample, dtype=np.uint8)
best = np.zeros(nsample, dtype=np.float32)
emu.lib.emu_phase1.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_uint64] + [ctypes.c_void_p] * 2
emu.lib.emu_phase1(emu.h, qs.ctypes.data, nsample, cls.ctypes.data, best.ctypes.data)
depth = emu.lib.emu_max_depth(emu.h)
print(f"cloud {cloud}, {nsample} queries, tree depth {depth}")
print("class  share    mean cost   p99    max")
for c in range(8):
    m = cls == c
    if m.any():
        print(f"{c:5d} {m.mean():7.4f} {cost[m].mean():10.1f} {np.percentile(cost[m], 99):6.0f} {cost[m].max():6d}")
for thr in (200, 400, 800, 1600):
    heavy = cost > thr
    print(f"cost > {thr}: {heavy.sum()} queries ({heavy.mean():.5f}); of them class 7: {(cls[heavy] == 7).mean():.3f}, "
          f"class >= 5: {(cls[heavy] >= 5).mean():.3f}")
print("total cost share of class 7:", cost[cls == 7].sum() / cost.sum())
np.savez("/tmp/cost_%s.npz" % cloud, cost=cost, cls=cls, best=best, cnt=cnt, sel=sel)
