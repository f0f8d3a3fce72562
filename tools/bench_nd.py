#!/usr/bin/env python3
"""Any-dimension kernels (dim > 3): knn and radius throughput on uniform clouds, device-resident.
One JSON line per dimension."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds
    for dim, n, nq, radius in ((4, 2_000_000, 1_000_000, 0.0004), (8, 1_000_000, 500_000, 0.04), (16, 500_000, 200_000, 0.5)):
        pts, q = ds.uniform_cloud(n, dim, 1), ds.uniform_cloud(nq, dim, 2)
        tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
        dq = torch.from_numpy(q).cuda()
        res = {"dim": dim, "n": n, "nq": nq}
        for k in (1, 16):
            out = torch.empty((nq, k, 2), dtype=torch.int32, device="cuda")
            tree.search_knn(dq, k, out); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): tree.search_knn(dq, k, out)
            torch.cuda.synchronize()
            res[f"knn{k}_Mq_s"] = round(nq / ((time.perf_counter() - t0) / 3) / 1e6, 1)
        off, raw = tree.search_radius_device(dq, radius); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): off, raw = tree.search_radius_device(dq, radius)
        torch.cuda.synchronize()
        res["radius_Mq_s"] = round(nq / ((time.perf_counter() - t0) / 3) / 1e6, 1)
        res["radius_hits_per_query"] = round(int(off[-1].item()) / nq, 1)
        print(json.dumps(res), flush=True)
        tree.close()

if __name__ == "__main__":
    main()
