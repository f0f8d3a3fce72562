#!/usr/bin/env python3
"""Experiment: successive batches of BASELINE config 2 issued on S HIP streams (one tree replica and one
scratch block per stream), so that the long, mostly idle tail of one batch's phase 2 overlaps the sort and
phase 1 of the next.  Prints steps/s per S."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds
    pts, q = ds.config2_clouds("L", ds.CONFIG2_N, ds.CONFIG2_NQ)
    nq = len(q)
    dq = torch.from_numpy(q).cuda()
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for S in (1, 2, 3, 4):
        trees = [pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0) for _ in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)]
        outs = [torch.empty((nq, k, 2), dtype=torch.int32, device="cuda") for _ in range(S)]
        torch.cuda.synchronize()

        def run(steps):
            for i in range(steps):
                s = i % S
                with torch.cuda.stream(streams[s]):
                    trees[s].search_knn(dq, k, outs[s])
            torch.cuda.synchronize()
        run(2 * S)
        t0 = time.perf_counter(); steps = 24; run(steps); dt = time.perf_counter() - t0
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        print(json.dumps({"streams": S, "k": k, "ms_per_step": round(dt / steps * 1e3, 3),
                          "Mq_s": round(nq * steps / dt / 1e6, 1), "replicas_agree": same}), flush=True)
        for t in trees: t.close()


if __name__ == "__main__":
    main()
