#!/usr/bin/env python3
"""Step time of the double k-NN search over caps (test hook knn64_cap; 0 = uncapped) and batch sizes on BASELINE
config 2's cloud L in float64: what the rule of knn64_cap (ptk_backend_f64.hpp) is fitted to.  One JSON line per (k, nq)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds
    ks = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 4, 16, 32]
    caps = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 4, 8, 16, 32, 64, 128, 256, 512]
    p, q = ds.config2_clouds("L")
    tree = pt.KdTree(p.astype(np.float64), pt.Metric.L2Squared, 10, device=0)
    for k in ks:
        for nq in (20_000, 150_000, 900_000, 3_600_000, len(q)):
            qq = np.ascontiguousarray(q[:: len(q) // nq][:nq].astype(np.float64))
            dq = torch.from_numpy(qq).cuda()
            out = torch.zeros((len(qq), k, 2), dtype=torch.int64, device="cuda")
            row = {"k": k, "nq": len(qq)}
            for cap in caps:
                pt.set_test_knobs(knn64_cap=cap)
                tree.search_knn(dq, k, out); torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter(); tree.search_knn(dq, k, out); torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                c = tree.knn_coop_counts()
                row[str(cap)] = [round(sorted(ts)[1], 3), c["cooperative"], c["redone"]]
            pt.set_test_knobs(knn64_cap=None)
            print(json.dumps(row), flush=True)
            del out, dq


if __name__ == "__main__":
    main()
