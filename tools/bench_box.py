#!/usr/bin/env python3
"""search_box throughput on BASELINE config 2's clouds (host buffers in, ragged rows out: the C ABI's
ptk_search_box), boxes of half-width h around the queries.  One JSON line per cloud."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import pico_tree_amd as pt
    from pico_tree_amd import datasets as ds
    for cloud, half in (("L", 0.5), ("U", 1.0)):
        pts, q = ds.config2_clouds(cloud)
        q = q[:2_000_000]
        tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
        boxes = np.empty((2 * len(q), 3), dtype=np.float32)
        boxes[0::2], boxes[1::2] = q - np.float32(half), q + np.float32(half)
        tree.profile(enable=True, reset=True)
        r = tree.search_box(boxes)
        tree.profile(reset=True)
        t0 = time.perf_counter()
        r = tree.search_box(boxes)
        dt = time.perf_counter() - t0
        prof = tree.profile()
        print(json.dumps({"cloud": cloud, "boxes": len(q), "half_width": half, "hits_per_box": round(len(r.flat) / len(q), 1),
                          "Mboxes_s_end_to_end": round(len(q) / dt / 1e6, 1), "ms": round(dt * 1e3, 1),
                          "kernel_ms": round(prof["search_ms"], 2), "other_ms": round(prof["other_ms"], 2)}), flush=True)
        import torch
        dmn, dmx = torch.from_numpy(q - np.float32(half)).cuda(), torch.from_numpy(q + np.float32(half)).cuda()
        tree.search_box_device(dmn, dmx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        off, rows = tree.search_box_device(dmn, dmx)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"cloud": cloud, "form": "device buffers", "Mboxes_s": round(len(q) / dt / 1e6, 1),
                          "ms": round(dt * 1e3, 2), "rows_equal": bool(np.array_equal(rows.cpu().numpy(), r.flat))}), flush=True)
        tree.close()

if __name__ == "__main__":
    main()
