#!/usr/bin/env python3
"""Does a double-precision k-NN batch wait for its longest query?  Step time over batch sizes on BASELINE config 2's
cloud L in float64 (every s-th query of the generated order, device-resident): a floor that does not shrink with the
batch is the longest search of the cloud, which only a cap + cooperative finish takes away."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds
    p, q = ds.config2_clouds("L")
    for dtype in (np.float64, np.float32):
        tree = pt.KdTree(p.astype(dtype), pt.Metric.L2Squared, 10, device=0)
        for k in (1, 16):
            row = {"dtype": np.dtype(dtype).name, "k": k}
            for nq in (20_000, 150_000, 900_000, 3_600_000, len(q)):
                qq = np.ascontiguousarray(q[:: len(q) // nq][:nq].astype(dtype))
                dq = torch.from_numpy(qq).cuda()
                out = torch.zeros((len(qq), k, 2), dtype=torch.int64 if dtype is np.float64 else torch.int32, device="cuda")
                tree.search_knn(dq, k, out); torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter(); tree.search_knn(dq, k, out); torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                row[str(len(qq))] = round(sorted(ts)[2], 3)
                del out, dq
            print(json.dumps(row), flush=True)
        tree.close()


if __name__ == "__main__":
    main()
