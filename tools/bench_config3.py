#!/usr/bin/env python3
"""BASELINE config 3: same LiDAR-like clouds as config 2, knn = 16 and search_radius r = 1.0
(metric units, i.e. squared radius 1.0), one MI355X.  Queries and tree resident in HBM.
Prints one JSON line per search with Mqueries/s, the kernel time (HIP events inside libptk)
and the algorithmic-bytes roofline fraction (visit counters from the oracle on a sample)."""
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
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    dq = torch.from_numpy(q).cuda()
    ref = oracle.Oracle(pts, 10, "port")
    ref.set_threads(ref.max_threads())
    rng = np.random.default_rng(7)
    sample = np.sort(rng.choice(nq, 100_000, replace=False))

    # ---- knn = 16 ----
    k = 16
    out = torch.empty((nq, k, 2), dtype=torch.int32, device="cuda")
    tree.search_knn(dq, k, out)
    torch.cuda.synchronize()
    tree.profile(enable=True, reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        tree.search_knn(dq, k, out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    prof = tree.profile(enable=False, reset=True)
    want, cnt = ref.search_knn(q[sample], k, counters=True)
    got = pt.DeviceNeighbors(out).numpy()[sample]
    m = cnt.astype(np.float64).mean(0)
    b = 12 + 8 * k + 16 * m[0] + 8 * m[1] + 16 * m[2]
    kernel_ms = prof["search_ms"] / max(prof["launches"], 1)
    print(json.dumps({"search": f"knn={k}", "cloud": cloud, "Mq_s": round(nq / ms / 1e3, 1), "ms_per_step": round(ms, 3),
                      "kernel_ms": round(kernel_ms, 3), "reorder_ms": round(prof["reorder_ms"] / steps, 3),
                      "parity_sample_ok": bool(got.tobytes() == want.tobytes()),
                      "bytes_per_query": round(b, 1), "visits": [round(x, 2) for x in m[:3]],
                      "roofline_frac": round(b * nq / (kernel_ms * 1e-3) / 8e12, 4)}), flush=True)
    del out

    # ---- radius ----
    radius = 1.0
    off, raw = tree.search_radius_device(dq, radius)   # warm-up
    torch.cuda.synchronize()
    tree.profile(enable=True, reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        off, raw = tree.search_radius_device(dq, radius)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    prof = tree.profile(enable=False, reset=True)
    total = int(off[-1].item())
    o2, flat = ref.search_radius(q[sample[:20000]], radius)
    offs = off.cpu().numpy()
    ok = True
    rawn = raw.cpu().numpy()
    for j, qi in enumerate(sample[:20000][::97]):
        jj = j * 97
        a = rawn[offs[qi]:offs[qi + 1]].view(pt.NEIGHBOR)[:, 0]
        bref = flat[int(o2[jj]):int(o2[jj + 1])]
        ok = ok and a.tobytes() == bref.tobytes()
    kernel_ms = prof["search_ms"] / steps
    print(json.dumps({"search": f"radius r2={radius}", "cloud": cloud, "Mq_s": round(nq / ms / 1e3, 1),
                      "ms_per_step": round(ms, 3), "kernel_ms_count_plus_fill": round(kernel_ms, 3),
                      "reorder_ms": round(prof["reorder_ms"] / steps, 3), "hits_per_query": round(total / nq, 2),
                      "parity_sample_ok": bool(ok)}), flush=True)


if __name__ == "__main__":
    main()
