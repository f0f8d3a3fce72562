"""rocprofv3 target: the double k-NN search of BASELINE config 2's cloud L at chosen (k, cap) pairs (test hook knn64_cap).
    rocprofv3 --kernel-trace --stats -d out -o p64 -- python tools/prof_knn64.py 1:128 4:128 16:256 16:0"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
p, q = ds.config2_clouds("L")
tree = pt.KdTree(p.astype(np.float64), pt.Metric.L2Squared, 10, device=0)
dq = torch.from_numpy(q.astype(np.float64)).cuda()
for spec in sys.argv[1:] or ["1:128", "16:256"]:
    k, cap = (int(x) for x in spec.split(":"))
    out = torch.zeros((len(q), k, 2), dtype=torch.int64, device="cuda")
    pt.set_test_knobs(knn64_cap=cap)
    for _ in range(3):
        tree.search_knn(dq, k, out)
    torch.cuda.synchronize()
    del out
