#!/usr/bin/env python3
"""Throughput of the metric_l1 / metric_lpinf searches (ptk_tree_set_metric) on the BASELINE
config 2 clouds, beside L2 squared on the same kernels' default route.  One JSON line per search;
parity against the oracle on a sample of rows."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import oracle
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds

    cloud = sys.argv[1] if len(sys.argv) > 1 else "L"
    steps = 5
    pts, q = ds.config2_clouds(cloud)
    nq = len(q)
    dq = torch.from_numpy(q).cuda()
    sample = np.sort(np.random.default_rng(7).choice(nq, 50_000, replace=False))
    results = []
    for metric in ("L2Squared", "L1", "LPInf"):   # all GPU timings first, parity afterwards
        tree = pt.KdTree(pts, pt.Metric[metric], 10, device=0)
        for k in (1, 16):
            out = torch.empty((nq, k, 2), dtype=torch.int32, device="cuda")
            for _ in range(2):
                tree.search_knn(dq, k, out)
            torch.cuda.synchronize()
            tree.profile(enable=True, reset=True)
            t0 = time.perf_counter()
            for _ in range(steps):
                tree.search_knn(dq, k, out)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            prof = tree.profile(enable=False, reset=True)
            results.append((metric, k, ms, prof["search_ms"] / max(prof["launches"], 1),
                            pt.DeviceNeighbors(out).numpy()[sample]))
            del out
        tree.close()
    for metric, k, ms, kernel_ms, got in results:
        ref = oracle.Oracle(pts, 10, "port", metric)
        ref.set_threads(ref.max_threads())
        want = ref.search_knn(q[sample], k)
        print(json.dumps({"metric": metric, "search": f"knn={k}", "cloud": cloud,
                          "Mq_s": round(nq / ms / 1e3, 1), "ms_per_step": round(ms, 3), "kernel_ms": round(kernel_ms, 3),
                          "parity_sample_ok": bool(got.reshape(want.shape).tobytes() == want.tobytes())}), flush=True)


if __name__ == "__main__":
    main()
