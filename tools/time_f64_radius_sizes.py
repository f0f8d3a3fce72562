"""Step time of the double radius search (count + fill through the host entry is PCIe-bound; here: the Python device path
if there is one, else the host entry) over batch sizes on BASELINE config 2's cloud L."""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
p, q = ds.config2_clouds("L")
for dtype in (np.float64, np.float32):
    tree = pt.KdTree(p.astype(dtype), pt.Metric.L2Squared, 10, device=0)
    row = {"dtype": np.dtype(dtype).name}
    for nq in (2000, 20000, 150000):
        qq = np.ascontiguousarray(q[:: len(q) // nq][:nq].astype(dtype))
        tree.search_radius(qq, 1.0)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); r = tree.search_radius(qq, 1.0); ts.append((time.perf_counter() - t0) * 1e3)
        row[str(nq)] = [round(sorted(ts)[1], 2), int(r.offsets[-1])]
    print(row, flush=True)
