#!/usr/bin/env python3
"""One shard of BASELINE configs[3] on this GPU (rows [0, ceil(nq / 8)) of the batch against the whole tree, k = 1), stepped
back to back: the loop tools/shard_prof.sh traces, and an A/B of environment knobs.

    python tools/shard_step.py [--configs "A=1;B=2"] [--rounds 30] [--parts 8]
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
    ap.add_argument("--configs", default="")
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--k", type=int, default=1)
    args = ap.parse_args()
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds

    pts, q = ds.config2_clouds("L")
    per = (len(q) + args.parts - 1) // args.parts
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    dq = torch.from_numpy(np.ascontiguousarray(q[:per])).cuda()
    out = torch.empty((per, args.k, 2), dtype=torch.int32, device="cuda")
    configs = [c.strip() for c in args.configs.split(";")]
    names = {kv.split("=")[0] for c in configs for kv in filter(None, c.split(","))}
    base = None
    res = {c: [] for c in configs}
    same = {}
    for rnd in range(4):
        for c in configs:
            for n in names:
                os.environ.pop(n, None)
            for kv in filter(None, c.split(",")):
                a, b = kv.split("=")
                os.environ[a] = b
            for _ in range(3):
                tree.search_knn(dq, args.k, out)
            torch.cuda.synchronize()
            if rnd == 0:
                rows = out.cpu().numpy().copy()
                if base is None:
                    base = rows
                same[c] = bool(np.array_equal(rows, base))
            t0 = time.perf_counter()
            for _ in range(args.rounds):
                tree.search_knn(dq, args.k, out)
            torch.cuda.synchronize()
            res[c].append((time.perf_counter() - t0) / args.rounds * 1e3)
    for c in configs:
        tree_counts = None
        print(c or "(default)", json.dumps({"queries": per, "ms_per_step": round(statistics.median(res[c]), 4),
                                            "min": round(min(res[c]), 4), "same_rows": same[c]}), flush=True)


if __name__ == "__main__":
    main()
