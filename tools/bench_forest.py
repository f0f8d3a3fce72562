#!/usr/bin/env python3
"""BASELINE config 5: approximate knn = 10 through the kd-forest on a SIFT-1M-shaped cloud.

Synthetic 1 000 000 x 128 float32 "descriptors" (datasets.sift_like_cloud; the real SIFT-1M
cannot be downloaded here) + 10 000 queries; forest of 8 trees, max_leaf_size 32, 64 leaves
visited per tree (the reference's SIFT setting, examples/kd_forest/kd_forest.cpp:118-123), k = 10.
Reports queries/s (queries and forest resident in HBM), recall@1 / recall@10 against the exact
answer (brute force on the GPU in float64-checked float32), and -- where oracle/_ref was built --
the reference kd_forest on the host cores next to it (a reported baseline, not the target).

    python tools/bench_forest.py [--n 1000000] [--nq 10000] [--trees 8] [--leaf 32] [--leaves 64] [--k 10]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def exact_knn(pts_d, q_d, k, chunk=500):
    """Brute-force top-k on the device (float32 expansion, candidates re-ranked exactly)."""
    import torch
    out = []
    pn = (pts_d * pts_d).sum(1)
    for s in range(0, q_d.shape[0], chunk):
        qq = q_d[s:s + chunk]
        d = pn[None, :] - 2.0 * (qq @ pts_d.T)
        cand = torch.topk(d, k + 8, dim=1, largest=False).indices          # a few spare for rounding
        diff = pts_d[cand].double() - qq[:, None, :].double()
        dd = (diff * diff).sum(2)
        order = torch.argsort(dd, dim=1)[:, :k]
        out.append(torch.gather(cand, 1, order).cpu().numpy())
    return np.concatenate(out)


def recall(found_idx, exact_idx, k):
    return float(np.mean([len(set(f[:k].tolist()) & set(e[:k].tolist())) / k for f, e in zip(found_idx, exact_idx)]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--trees", type=int, default=8)
    ap.add_argument("--leaf", type=int, default=32)
    ap.add_argument("--leaves", type=int, default=64)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu-queries", type=int, default=2000)
    args = ap.parse_args()
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds

    t0 = time.perf_counter()
    pts = ds.sift_like_cloud(args.n, args.dim, seed=1)
    q = ds.sift_like_cloud(args.nq, args.dim, seed=2)
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    forest = pt.KdForest(pts, args.leaf, args.trees, seed=1, device=0)
    build_s = time.perf_counter() - t0
    dq = torch.from_numpy(q).cuda()
    res = forest.search_knn(dq, args.k, args.leaves)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = forest.search_knn(dq, args.k, args.leaves)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    got = res.numpy()
    exact = exact_knn(torch.from_numpy(pts).cuda(), dq, args.k)
    out = {"metric": f"queries/s, kd_forest approximate knn={args.k}, {args.trees} trees, leaf {args.leaf}, "
                     f"{args.leaves} leaves/tree, {args.n} x {args.dim} float32",
           "value": round(args.nq / ms * 1e3, 1), "unit": "queries/s", "ms_per_batch": round(ms, 3),
           "recall_at_1": round(recall(got["index"].reshape(args.nq, -1), exact, 1), 4),
           f"recall_at_{args.k}": round(recall(got["index"].reshape(args.nq, -1), exact, args.k), 4),
           "data": "synthetic (mixture of 1000 Gaussians, SIFT-like range)", "gen_s": round(gen_s, 1),
           "host_build_upload_s": round(build_s, 1),
           "leaf_bytes_read_per_query": args.trees * args.leaves * args.leaf * args.dim * 4,
           "queue_entries_dropped": forest.dropped}
    out["hbm_gbs_leaf_scans"] = round(out["leaf_bytes_read_per_query"] * args.nq / (ms * 1e-3) / 1e9, 1)
    import oracle
    if oracle.have_reference_forest():
        nq_cpu = min(args.cpu_queries, args.nq)
        t0 = time.perf_counter()
        ref = oracle.ReferenceForest(pts, args.leaf, args.trees)
        ref_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        r = ref.search_knn(q[:nq_cpu], 1, args.leaves)
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(nq_cpu / cpu_s, 1), "unit": "queries/s", "kind": "reference",
                               "what": "reference kd_forest::search_nn, OpenMP over queries",
                               "cores": len(os.sched_getaffinity(0)), "sample": f"first {nq_cpu} queries",
                               "recall_at_1": round(recall(r["index"].reshape(nq_cpu, -1), exact[:nq_cpu], 1), 4),
                               "build_s": round(ref_build, 1)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
