#!/usr/bin/env python3
"""Within-process A/B of environment-selected forms of the k-NN search.

    python tools/ab_env.py --configs "PTK_TEST_KNOBS=p2_cap=0;PTK_TEST_KNOBS=p2_cap=16;PTK_TEST_KNOBS=p2_cap=64" --rounds 5 [--cloud L] [--k 1]

Each ';'-separated config is a ','-separated list of NAME=VALUE settings (libptk reads its knobs
at every call).  The configs are interleaved over several rounds in ONE process on the same
resident data; per config: median traversal-kernel time (HIP events inside libptk), the other
kernels, the end-to-end step time, and whether the rows equal those of the first config.
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
    ap.add_argument("--configs", required=True)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--cloud", default="L")
    ap.add_argument("--order", default="generated")
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--nq", type=int, default=None)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch

    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds

    if os.environ.get("PTK_LIBRARY"):  # (tools/ab_libs.sh: a library of an older commit may lack the newest entry points)
        import ctypes
        probe = ctypes.CDLL(os.environ["PTK_LIBRARY"])
        pt._SIGNATURES = {k: v for k, v in pt._SIGNATURES.items() if hasattr(probe, k)}
    n = args.n or ds.CONFIG2_N
    nq = args.nq or ds.CONFIG2_NQ
    pts, q = ds.config2_clouds(args.cloud, n, nq)
    if args.order == "morton":
        q = np.ascontiguousarray(q[ds.morton_order(q)])
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    dq = torch.from_numpy(q).cuda()
    out = torch.empty((nq, args.k, 2), dtype=torch.int32, device="cuda")
    configs = [c.strip() for c in args.configs.split(";")]
    names = set()
    for c in configs:
        for kv in filter(None, c.split(",")):
            names.add(kv.split("=")[0])
    base = None
    stats = {c: {"kernel_ms": [], "step_ms": [], "reorder_ms": [], "other_ms": []} for c in configs}
    ok, counts = {}, {}
    for rnd in range(args.rounds + 1):  # round 0 = warm-up + correctness
        for c in configs:
            for nme in names:
                os.environ.pop(nme, None)
            for kv in filter(None, c.split(",")):
                a, b = kv.split("=")
                os.environ[a] = b
            tree.profile(enable=True, reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tree.search_knn(dq, args.k, out)
            torch.cuda.synchronize()
            step = (time.perf_counter() - t0) * 1e3
            p = tree.profile(enable=False, reset=True)
            if rnd == 0:
                res = out.cpu().numpy().copy()
                if base is None:
                    base = res
                ok[c] = bool(np.array_equal(res, base))
                if True:  # (k > 1: the capped general kernel leaves its counts in the same words)
                    try:
                        counts[c] = tree.knn1_counts() if args.k == 1 else tree.knn_coop_counts()
                    except Exception as exc:  # noqa: BLE001
                        counts[c] = str(exc)
            else:
                stats[c]["kernel_ms"].append(p["search_ms"])
                stats[c]["reorder_ms"].append(p["reorder_ms"])
                stats[c]["other_ms"].append(p["other_ms"])
                stats[c]["step_ms"].append(step)
    report = {}
    for c in configs:
        s = stats[c]
        report[c] = {"kernel_ms": round(statistics.median(s["kernel_ms"]), 4),
                     "kernel_ms_min": round(min(s["kernel_ms"]), 4),
                     "other_ms": round(statistics.median(s["other_ms"]), 4),
                     "reorder_ms": round(statistics.median(s["reorder_ms"]), 4),
                     "step_ms": round(statistics.median(s["step_ms"]), 4),
                     "Mq_s_step": round(nq / statistics.median(s["step_ms"]) / 1e3, 1),
                     "same_as_first": ok[c], "counts": counts.get(c)}
        print(c or "(default)", json.dumps(report[c]), flush=True)
    line = json.dumps({"cloud": args.cloud, "order": args.order, "k": args.k, "n": n, "nq": nq, "report": report})
    print(line)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
