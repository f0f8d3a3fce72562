#!/usr/bin/env python3
"""Double-precision kernels: knn throughput on BASELINE config 2's cloud L converted to float64 (queries in
generated order and Morton-presorted by the caller -- the f64 path does not reorder) and on uniform clouds of
higher dimension, device-resident.  One JSON line per case; the float32 tree on the same data for context."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=3):
    import torch
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def morton_order(q, lo, hi):
    g = np.clip(((q - lo) / (hi - lo) * 1024).astype(np.int64), 0, 1023)
    def spread(x):
        x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249
        return x
    key = spread(g[:, 0]) | (spread(g[:, 1]) << 1) | (spread(g[:, 2]) << 2)
    return np.argsort(key, kind="stable")


def main():
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds
    small = "--small" in sys.argv
    n, nq = (1_000_000, 1_000_000) if small else (7_733_372, 7_200_863)
    p32 = ds.lidar_cloud(n, seed=1); q32 = ds.lidar_cloud(nq, seed=2, pose=(3.0, 1.5))
    order = morton_order(q32, p32.min(0), p32.max(0))
    for dtype in (np.float64, np.float32):
        pts, q = p32.astype(dtype), q32.astype(dtype)
        t0 = time.perf_counter(); tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0); build = time.perf_counter() - t0
        for label, qq in (("generated", q), ("morton", np.ascontiguousarray(q[order]))):
            dq = torch.from_numpy(qq).cuda()
            res = {"cloud": "L", "dtype": np.dtype(dtype).name, "n": n, "nq": nq, "order": label, "create_s": round(build, 2),
                   "depth": tree.info()["max_depth"]}
            for k in (1, 16):
                out = torch.zeros((nq, k, 2), dtype=torch.int64 if dtype is np.float64 else torch.int32, device="cuda")
                res[f"knn{k}_Mq_s"] = round(nq / timed(lambda: tree.search_knn(dq, k, out)) / 1e6, 1)
                del out
            print(json.dumps(res), flush=True)
        tree.close()
    for dim, n, nq in ((8, 1_000_000, 500_000),):
        pts, q = ds.uniform_cloud(n, dim, 1).astype(np.float64), ds.uniform_cloud(nq, dim, 2).astype(np.float64)
        tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
        dq = torch.from_numpy(q).cuda()
        res = {"cloud": "U", "dtype": "float64", "dim": dim, "n": n, "nq": nq}
        for k in (1, 16):
            out = torch.zeros((nq, k, 2), dtype=torch.int64, device="cuda")
            res[f"knn{k}_Mq_s"] = round(nq / timed(lambda: tree.search_knn(dq, k, out)) / 1e6, 1)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
