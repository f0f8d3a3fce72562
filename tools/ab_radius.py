#!/usr/bin/env python3
"""Within-process A/B of environment-selected forms of the radius search on BASELINE config 3 (cloud L, r^2 = 1):
    python tools/ab_radius.py --configs "PTK_TEST_KNOBS=radius_lists=0;PTK_TEST_KNOBS=radius_lists=1" [--rounds 3]
Per config: median kernel ms (count pass + scan + fill, HIP events inside libptk), step ms, rows equal to the first."""
import argparse, json, os, statistics, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", required=True)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--cloud", default="L")
    ap.add_argument("--radius", type=float, default=1.0)
    ap.add_argument("--nq", type=int, default=None, help="only the first nq queries of the batch")
    args = ap.parse_args()
    import torch
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds
    pts, q = ds.config2_clouds(args.cloud)
    if args.nq:
        q = np.ascontiguousarray(q[:args.nq])
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    dq = torch.from_numpy(q).cuda()
    configs = [c.strip() for c in args.configs.split(";")]
    names = {kv.split("=")[0] for c in configs for kv in filter(None, c.split(","))}
    stats = {c: {"kernel_ms": [], "step_ms": []} for c in configs}
    base, same = None, {}
    for rnd in range(args.rounds + 1):
        for c in configs:
            for n in names:
                os.environ.pop(n, None)
            for kv in filter(None, c.split(",")):
                a, b = kv.split("=")
                os.environ[a] = b
            off = raw = None
            tree.profile(enable=True, reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            off, raw = tree.search_radius_device(dq, args.radius)
            torch.cuda.synchronize()
            step = (time.perf_counter() - t0) * 1e3
            p = tree.profile(enable=False, reset=True)
            if rnd == 0:
                import hashlib
                h = hashlib.sha256(raw.cpu().numpy().tobytes()).hexdigest() + hashlib.sha256(off.cpu().numpy().tobytes()).hexdigest()
                if base is None:
                    base = h
                same[c] = h == base
            else:
                stats[c]["kernel_ms"].append(p["search_ms"])
                stats[c]["step_ms"].append(step)
            del off, raw
    for c in configs:
        print(c or "(default)", json.dumps({"kernel_ms": round(statistics.median(stats[c]["kernel_ms"]), 3),
                                            "step_ms": round(statistics.median(stats[c]["step_ms"]), 3),
                                            "same_as_first": same[c]}), flush=True)

if __name__ == "__main__":
    main()
