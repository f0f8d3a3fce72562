import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
p, q = ds.config2_clouds("L")
tree = pt.KdTree(p.astype(np.float64), pt.Metric.L2Squared, 10, device=0)
for k, nq, cap in ((32, 150000, 32), (32, 900000, 64), (24, 150000, 32), (16, 900000, 64)):
    qq = np.ascontiguousarray(q[:: len(q) // nq][:nq].astype(np.float64))
    dq = torch.from_numpy(qq).cuda()
    out = torch.zeros((len(qq), k, 2), dtype=torch.int64, device="cuda")
    pt.set_test_knobs(knn64_cap=cap)
    tree.search_knn(dq, k, out); torch.cuda.synchronize()
    print(k, nq, cap, tree.knn_coop_counts(), flush=True)
