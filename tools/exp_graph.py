#!/usr/bin/env python3
"""Experiment: one k = 1 search captured in a HIP graph (through torch.cuda.CUDAGraph) and replayed, against the
same search launched kernel by kernel.  python tools/exp_graph.py [--nq N]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=None)
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    import torch

    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds

    nq = args.nq or ds.CONFIG2_NQ
    pts, q = ds.config2_clouds("L", ds.CONFIG2_N, nq)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    dq = torch.from_numpy(q).cuda()
    out = torch.empty((nq, 1, 2), dtype=torch.int32, device="cuda")
    ref = torch.empty_like(out)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            tree.search_knn(dq, 1, ref)
    torch.cuda.synchronize()

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    def plain():
        with torch.cuda.stream(s):
            tree.search_knn(dq, 1, out)

    ms_plain = timed(plain)
    tree.profile(enable=True, reset=True)
    ms_prof = timed(plain)
    tree.profile(enable=False)
    ms_plain2 = timed(plain)
    print(f"nq {nq}: plain {ms_plain:.4f} / {ms_plain2:.4f} ms, with the profiling events {ms_prof:.4f} ms")
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            tree.search_knn(dq, 1, out)
    except Exception as exc:  # noqa: BLE001
        print("capture failed:", repr(exc)[:300])
        return
    out.zero_()
    ms_graph = timed(g.replay)
    print(f"nq {nq}: plain {ms_plain:.4f} ms, graph {ms_graph:.4f} ms, rows equal: {bool(torch.equal(out, ref))}")


if __name__ == "__main__":
    main()
