#!/usr/bin/env python3
"""Times the host-buffer entry (numpy in, numpy out: H2D + search + D2H per call) beside the
device-resident one on BASELINE config 2.  One JSON line."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds
    cloud = sys.argv[1] if len(sys.argv) > 1 else "L"
    pts, q = ds.config2_clouds(cloud)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    res = {"cloud": cloud}
    for k in (1, 16):
        out = np.empty((len(q), k) if k > 1 else (len(q),), dtype=pt.NEIGHBOR)
        tree.search_knn(q, k, out)
        t0 = time.perf_counter()
        for _ in range(5):
            tree.search_knn(q, k, out)
        host_ms = (time.perf_counter() - t0) / 5 * 1e3
        dq = torch.from_numpy(q).cuda()
        dout = torch.empty((len(q), k, 2), dtype=torch.int32, device="cuda")
        tree.search_knn(dq, k, dout)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tree.search_knn(dq, k, dout)
        torch.cuda.synchronize()
        dev_ms = (time.perf_counter() - t0) / 5 * 1e3
        same = bool(np.array_equal(out.view(np.int32).reshape(-1), dout.cpu().numpy().reshape(-1)))
        res[f"knn{k}"] = {"host_buffers_ms": round(host_ms, 3), "device_resident_ms": round(dev_ms, 3),
                          "bytes_moved_MB": round((q.nbytes + out.nbytes) / 1e6, 1), "same_result": same}
    print(json.dumps(res))

if __name__ == "__main__":
    main()
