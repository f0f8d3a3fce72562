#!/usr/bin/env python3
"""Within-process A/B of the host-buffer entry (ptk_search_knn on numpy arrays) under environment settings.
python tools/ab_host.py --configs "PTK_TEST_KNOBS=host_piece=900108;PTK_TEST_KNOBS=host_piece=1800216" [--k 1] [--rounds 5]"""
import argparse, json, os, statistics, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", required=True)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--cloud", default="L")
    args = ap.parse_args()
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds
    pts, q = ds.config2_clouds(args.cloud)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    out = np.empty((len(q), args.k) if args.k > 1 else (len(q),), dtype=pt.NEIGHBOR)
    configs = [c.strip() for c in args.configs.split(";")]
    names = {kv.split("=")[0] for c in configs for kv in filter(None, c.split(","))}
    times = {c: [] for c in configs}
    base = None
    for rnd in range(args.rounds + 1):
        for c in configs:
            for n in names:
                os.environ.pop(n, None)
            for kv in filter(None, c.split(",")):
                a, b = kv.split("=")
                os.environ[a] = b
            t0 = time.perf_counter()
            tree.search_knn(q, args.k, out)
            dt = (time.perf_counter() - t0) * 1e3
            if rnd == 0:
                if base is None:
                    base = out.copy()
                assert out.tobytes() == base.tobytes(), c
            else:
                times[c].append(dt)
    for c in configs:
        print(c or "(default)", json.dumps({"ms": round(statistics.median(times[c]), 3), "min": round(min(times[c]), 3)}), flush=True)

if __name__ == "__main__":
    main()
