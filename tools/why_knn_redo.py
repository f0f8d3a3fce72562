"""Why does the cooperative k > 1 search send queries to the redo list?  (ptk_debug_knn_coop_counts per k and batch size,
BASELINE config 2's cloud L, float32, the rule's own caps.)"""
import sys, os, numpy as np, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
p, q = ds.config2_clouds("L")
tree = pt.KdTree(p, pt.Metric.L2Squared, 10, device=0)
for k in (8, 16, 24, 32, 40, 56):
    for nq in (20000, 150000, 900000, len(q)):
        qq = np.ascontiguousarray(q[:: len(q) // nq][:nq])
        dq = torch.from_numpy(qq).cuda()
        out = torch.zeros((len(qq), k, 2), dtype=torch.int32, device="cuda")
        tree.search_knn(dq, k, out); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): tree.search_knn(dq, k, out)
        torch.cuda.synchronize()
        print(k, len(qq), round((time.perf_counter() - t0) / 3 * 1e3, 3), tree.knn_coop_counts(), flush=True)
