#!/usr/bin/env python3
"""Within-process A/B of the compiled k = 1 kernel geometries (PTK_KNN1_VARIANT).

Interleaves the variants over several rounds in ONE process on the same resident
data and reports, per variant, the median / min traversal-kernel time (HIP events
inside libptk) and the end-to-end step time.  Every variant's output is compared
with variant 0's (bit-exact) so a fast-but-wrong geometry cannot slip through.

    python tools/ab_knn1.py --variants 0,22,4 --rounds 5 [--cloud L|U] [--k 1]
"""

from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,22,4")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--cloud", default="L")
    ap.add_argument("--order", default="generated")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--nq", type=int, default=None)
    ap.add_argument("--env", default="PTK_KNN1_VARIANT")
    args = ap.parse_args()

    import torch

    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds

    n = args.n or ds.CONFIG2_N
    nq = args.nq or ds.CONFIG2_NQ
    pts, q = ds.config2_clouds(args.cloud, n, nq)
    if args.order == "morton":
        q = np.ascontiguousarray(q[ds.morton_order(q)])
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    dq = torch.from_numpy(q).cuda()
    out = torch.empty((nq, 1, 2), dtype=torch.int32, device="cuda")
    variants = [int(v) for v in args.variants.split(",")]
    base = None
    stats = {v: {"kernel_ms": [], "step_ms": [], "reorder_ms": []} for v in variants}
    ok = {}
    for rnd in range(args.rounds + 1):  # round 0 = warm-up + correctness
        for v in variants:
            os.environ[args.env] = str(v)
            tree.profile(enable=True, reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tree.search_knn(dq, 1, out)
            torch.cuda.synchronize()
            step = (time.perf_counter() - t0) * 1e3
            p = tree.profile(enable=False, reset=True)
            if rnd == 0:
                res = out.cpu().numpy().copy()
                if base is None:
                    base = res
                ok[v] = bool(np.array_equal(res, base))
            else:
                stats[v]["kernel_ms"].append(p["search_ms"])
                stats[v]["reorder_ms"].append(p["reorder_ms"])
                stats[v]["step_ms"].append(step)
    report = {}
    for v in variants:
        s = stats[v]
        report[v] = {"kernel_ms_median": round(statistics.median(s["kernel_ms"]), 4),
                     "kernel_ms_min": round(min(s["kernel_ms"]), 4),
                     "step_ms_median": round(statistics.median(s["step_ms"]), 4),
                     "reorder_ms_median": round(statistics.median(s["reorder_ms"]), 4),
                     "Mq_s_kernel": round(nq / statistics.median(s["kernel_ms"]) / 1e3, 1),
                     "same_as_first": ok[v]}
        print(v, json.dumps(report[v]), flush=True)
    print(json.dumps({"cloud": args.cloud, "order": args.order, "n": n, "nq": nq, "report": report}))


if __name__ == "__main__":
    main()
