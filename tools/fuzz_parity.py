#!/usr/bin/env python3
"""Randomised differential test on the GPU: libptk (through the Python host / C ABI) against the
oracle's CPU restatement on random trees, batches and search parameters.  Bit-exact comparison
(indices and float32 distance bits); radius rows in traversal order.  Prints one line per failing
case with everything needed to replay it (--seed S --case I), and a summary.

    python tools/fuzz_parity.py --cases 200 --seed 1
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


VERBOSE = False
DTYPE = np.float32  # --dtype float64: the double-precision entry points
LNINF = False       # --lninf: metric_lninf among the metrics drawn (changes what a seed draws: off by default)


def same_rows(got, want):
    """indices and distance bits (the float64 record has padding bytes that are not data)"""
    got = got.reshape(want.shape)
    return np.array_equal(got["index"], want["index"]) and \
        np.ascontiguousarray(got["distance"]).tobytes() == np.ascontiguousarray(want["distance"]).tobytes()


def make_cloud(rng, kind, n, dim):
    if kind == "uniform":
        p = rng.random((n, dim))
    elif kind == "clustered":
        c = rng.random((max(1, n // 200), dim))
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.01, (n, dim))
    elif kind == "lattice":
        p = np.round(rng.random((n, dim)) * 6) / 6
    elif kind == "duplicates":
        base = rng.random((max(1, n // 3), dim))
        p = base[rng.integers(0, len(base), n)]
    elif kind == "line":
        t = rng.random((n, 1))
        p = t * rng.random((1, dim)) + 0.25
    elif kind == "plane":
        p = rng.random((n, dim))
        p[:, -1] = 0.5
    else:
        raise ValueError(kind)
    return p


def one_case(rng, pt, oracle, torch, case):
    dim = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 7]))
    n = int(rng.choice([1, 2, 9, 100, 3000, 20000, 60000]))
    nq = int(rng.choice([1, 63, 64, 65, 1000, 5000]))
    leaf = int(rng.choice([1, 2, 5, 10, 16, 24]))
    kind = str(rng.choice(["uniform", "clustered", "lattice", "duplicates", "line", "plane"]))
    scale = float(rng.choice([1.0, 1.0, 1e-6, 1e6, 37.5]))
    shift = float(rng.choice([0.0, 0.0, -0.5, 100.0]))
    metric = str(rng.choice(["L2Squared", "L2Squared", "L1", "LPInf", "LNInf"] if LNINF else ["L2Squared", "L2Squared", "L1", "LPInf"]))
    pts = ((make_cloud(rng, kind, n, dim) + shift) * scale).astype(DTYPE)
    if rng.random() < 0.5:
        q = ((make_cloud(rng, kind, nq, dim) + shift) * scale).astype(DTYPE)
    else:  # queries near / on tree points
        q = pts[rng.integers(0, n, nq)] + (rng.normal(0, 1e-3, (nq, dim)) * scale * (rng.random() < 0.7)).astype(DTYPE)
        q = q.astype(DTYPE)
    desc = f"case {case}: dim {dim} n {n} nq {nq} leaf {leaf} {kind} scale {scale} shift {shift} {metric}"
    if VERBOSE:
        print(desc, flush=True)
    tree = pt.KdTree(pts, pt.Metric[metric], leaf, device=0)
    tree.set_reorder(int(rng.choice([pt.REORDER_AUTO, pt.REORDER_ON, pt.REORDER_OFF])))
    ref = oracle.Oracle(pts, leaf, "port", metric, dtype=DTYPE)
    bad = []
    for k in {1, int(rng.integers(1, min(n, 48) + 1)), min(n, int(rng.choice([2, 8, 33, 64])))}:
        e = float(rng.choice([1.0, 1.0, 1.3, 4.0]))
        want = ref.search_knn(q, k, e=None if e == 1.0 else e)
        got = tree.search_knn(q, k) if e == 1.0 else tree.search_knn(q, k, e)
        if not same_rows(got, want):
            bad.append(f"knn k={k} e={e}")
    nn = ref.search_knn(q, min(n, 4))["distance"][:, -1]
    radius = float(np.quantile(nn, rng.choice([0.1, 0.5, 0.9])) * rng.choice([1.0, 4.0])) or float(scale * 0.01)
    e = float(rng.choice([1.0, 1.0, 2.0]))
    off, flat = ref.search_radius(q, radius, e=None if e == 1.0 else e)
    got = tree.search_radius(q, radius) if e == 1.0 else tree.search_radius(q, radius, e)
    if not np.array_equal(got.offsets, off) or not same_rows(got.flat, flat):
        bad.append(f"radius r={radius} e={e}")
    if DTYPE is np.float32:
        dq = torch.from_numpy(q).cuda()
        doff, draw = tree.search_radius_device(dq, radius, e)
        if not np.array_equal(doff.cpu().numpy().astype(np.uint64), off) or draw.cpu().numpy().tobytes() != flat.tobytes():
            bad.append(f"radius (device buffers) r={radius} e={e}")
    else:  # device tensors in, device rows out
        kk = min(n, 3)
        dn = tree.search_knn(torch.from_numpy(q).cuda(), kk).numpy().reshape(nq, kk)
        if not same_rows(dn, ref.search_knn(q, kk)):
            bad.append("knn (device buffers)")
    if True:
        half = (rng.random((nq, dim)) * scale * (0.05 if dim <= 3 else 0.3)).astype(DTYPE)
        boxes = np.empty((2 * nq, dim), dtype=DTYPE)
        boxes[0::2], boxes[1::2] = q - half, q + half
        boff, bflat = ref.search_box(boxes[0::2].copy(), boxes[1::2].copy())
        b = tree.search_box(boxes)
        if not np.array_equal(b.offsets, boff) or not np.array_equal(b.flat, bflat):
            bad.append("box")
    tree.close()
    return desc, bad


def one_topological_case(rng, pt, oracle, torch, case):
    """metric_so2 (dim 1) / metric_se2_squared (dim 3) on coordinates in [0, 1], against the reference's own headers
    (oracle kind "reference": the restatement covers the euclidean metrics only): knn, radius, boxes through the seam."""
    metric = str(rng.choice(["SO2", "SE2Squared"]))
    dim = 1 if metric == "SO2" else 3
    n = int(rng.choice([2, 9, 100, 3000, 20000, 60000]))
    nq = int(rng.choice([1, 63, 64, 65, 1000, 5000]))
    leaf = int(rng.choice([1, 2, 5, 10, 16]))
    kind = str(rng.choice(["uniform", "clustered", "lattice", "duplicates"]))
    wrap01 = lambda a: np.ascontiguousarray(np.clip(a - np.floor(a), 0.0, 1.0).astype(DTYPE))
    pts = wrap01(make_cloud(rng, kind, n, dim))
    q = wrap01(make_cloud(rng, kind, nq, dim)) if rng.random() < 0.5 else \
        wrap01(pts[rng.integers(0, n, nq)] + rng.normal(0, 1e-3, (nq, dim)) * (rng.random() < 0.7))
    if nq > 4:  # next to the seam of the circle axis
        q[: nq // 4, -1] = (rng.random(nq // 4) * 1e-3).astype(DTYPE)
        q[nq // 4: nq // 2, -1] = (1.0 - rng.random(nq // 2 - nq // 4) * 1e-3).astype(DTYPE)
    desc = f"case {case}: {metric} n {n} nq {nq} leaf {leaf} {kind} {np.dtype(DTYPE).name}"
    if VERBOSE:
        print(desc, flush=True)
    tree = pt.KdTree(pts, pt.Metric[metric], leaf, device=0)
    ref = oracle.Oracle(pts, leaf, "reference", metric, dtype=DTYPE)
    bad = []
    for k in {1, int(rng.integers(1, min(n, 48) + 1)), min(n, int(rng.choice([2, 8, 33])))}:
        e = float(rng.choice([1.0, 1.0, 1.3]))
        want = ref.search_knn(q, k, e=None if e == 1.0 else e)
        got = tree.search_knn(q, k) if e == 1.0 else tree.search_knn(q, k, e)
        if not same_rows(got, want):
            bad.append(f"knn k={k} e={e}")
    nn = ref.search_knn(q, min(n, 4))["distance"][:, -1]
    radius = float(np.quantile(nn, rng.choice([0.1, 0.5, 0.9])) * rng.choice([1.0, 4.0])) or 1e-4
    off, flat = ref.search_radius(q, radius)
    got = tree.search_radius(q, radius)
    if not np.array_equal(got.offsets, off) or not same_rows(got.flat, flat):
        bad.append(f"radius r={radius}")
    half = (rng.random((nq, dim)) * 0.05).astype(DTYPE)
    mins, maxs = (q - half).astype(DTYPE), (q + half).astype(DTYPE)
    mins[:, -1] = np.where(mins[:, -1] < 0, mins[:, -1] + 1, mins[:, -1])  # an interval through the seam: min > max
    maxs[:, -1] = np.where(maxs[:, -1] > 1, maxs[:, -1] - 1, maxs[:, -1])
    boxes = np.empty((2 * nq, dim), dtype=DTYPE)
    boxes[0::2], boxes[1::2] = mins, maxs
    boff, bflat = ref.search_box(np.ascontiguousarray(mins), np.ascontiguousarray(maxs))
    b = tree.search_box(boxes)
    if not np.array_equal(b.offsets, boff) or not np.array_equal(b.flat, bflat):
        bad.append("box")
    tree.close()
    return desc, bad


def one_multi_case(rng, pt, oracle, torch, case):
    """ptk_multi_* with n replicas on the one GPU (PTK_MULTI_ALLOW_REPLICAS=1: the C library's row cutting, threads,
    streams, events and staging as on a node): random n, ragged and tiny batches, host and device buffers."""
    n_dev = int(rng.choice([1, 2, 3, 5, 8]))
    dim = int(rng.choice([1, 2, 3, 3]))
    n = int(rng.choice([9, 100, 3000, 20000]))
    nq = int(rng.choice([1, 2, 7, 63, 65, 1000, 5003]))
    leaf = int(rng.choice([1, 5, 10, 16]))
    kind = str(rng.choice(["uniform", "clustered", "lattice", "duplicates"]))
    pts = make_cloud(rng, kind, n, dim).astype(np.float32)
    q = make_cloud(rng, kind, nq, dim).astype(np.float32)
    desc = f"case {case}: multi x{n_dev} dim {dim} n {n} nq {nq} leaf {leaf} {kind}"
    if VERBOSE:
        print(desc, flush=True)
    multi = pt.MultiKdTree(pts, leaf, devices=[0] * n_dev)
    ref = oracle.Oracle(pts, leaf, "port")
    bad = []
    for k in {1, min(n, int(rng.choice([2, 8, 20])))}:
        e = float(rng.choice([1.0, 1.0, 1.5]))
        want = ref.search_knn(q, k, e=None if e == 1.0 else e)
        got = multi.search_knn(q, k) if e == 1.0 else multi.search_knn(q, k, e)
        if not same_rows(got, want):
            bad.append(f"knn (host buffers) k={k} e={e}")
        if dim <= 3 or True:
            rows = multi.search_knn(torch.from_numpy(q).cuda(), k) if e == 1.0 else multi.search_knn(torch.from_numpy(q).cuda(), k, e)
            rows = rows.numpy()
            torch.cuda.synchronize()
            if not same_rows(rows, want):
                bad.append(f"knn (device buffers) k={k} e={e}")
    nn = ref.search_knn(q, min(n, 4))["distance"][:, -1]
    radius = float(np.quantile(nn, 0.5) * rng.choice([1.0, 4.0])) or 0.01
    off, flat = ref.search_radius(q, radius)
    got = multi.search_radius(q, radius)
    if not np.array_equal(got.offsets, off) or not same_rows(got.flat, flat):
        bad.append(f"radius r={radius}")
    del multi
    return desc, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--case", type=int, default=-1, help="replay only this case")
    ap.add_argument("--verbose", action="store_true", help="print every case before it runs")
    ap.add_argument("--dtype", choices=["float32", "float64"], default="float32")
    ap.add_argument("--topological", action="store_true", help="metric_so2 / metric_se2_squared against the compiled reference")
    ap.add_argument("--lninf", action="store_true", help="metric_lninf among the metrics drawn")
    ap.add_argument("--multi", action="store_true", help="ptk_multi_* with replicas on the one GPU")
    args = ap.parse_args()
    global LNINF
    LNINF = args.lninf
    if args.multi:
        os.environ["PTK_MULTI_ALLOW_REPLICAS"] = "1"
    global DTYPE
    DTYPE = np.float64 if args.dtype == "float64" else np.float32
    import torch
    import oracle
    import pico_tree_amd as pt
    pt.allow_host_loop(True)  # (a search the device refuses -- a topological tree thousands of levels deep -- is part of what
                              # is fuzzed: served by the library's host loop, which must equal the oracle too; off by default)
    global VERBOSE
    VERBOSE = args.verbose
    failures = unsupported = 0
    for case in range(args.cases):
        rng = np.random.default_rng([args.seed, case])
        if args.case >= 0 and case != args.case:
            continue
        try:
            fn = one_topological_case if args.topological else (one_multi_case if args.multi else one_case)
            desc, bad = fn(rng, pt, oracle, torch, case)
        except pt.PtkError as err:
            # The one limit left: a point set so degenerate that the BUILD stops (deeper than 8192 levels: thousands of
            # coincident points peeling one level each; the reference's recursive builder overflows its stack on such
            # input, and so does the oracle).  A search the device refuses is served by the host loop -- not counted here.
            if "deeper than 8192 levels" not in str(err):
                raise
            unsupported += 1
            continue
        if bad:
            failures += 1
            print("FAIL", desc, "->", "; ".join(bad), flush=True)
    print(f"fuzz: {args.cases} cases, seed {args.seed}, {failures} failing, "
          f"{unsupported} not built (point set beyond the build limit of 8192 levels: the reference overflows its stack there)", flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
