import numpy as np, time, torch, sys
sys.path.insert(0,"/root/repo")
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
def timeit(tree, dq, k, raw64=False):
    tree.search_knn(dq,k); torch.cuda.synchronize()
    t0=time.perf_counter(); tree.search_knn(dq,k); torch.cuda.synchronize(); return round((time.perf_counter()-t0)*1e3,2)
pts,q=ds.config2_clouds("L", 2_000_000, 300_000)
t64=pt.KdTree(pts.astype(np.float64), pt.Metric.L2Squared, 10, device=0)
dq64=torch.from_numpy(q.astype(np.float64)).cuda()
print("f64 3-D 300k:", {k: timeit(t64,dq64,k) for k in (8,16,17,20,32,40)}, flush=True)
rng=np.random.default_rng(1)
p5=rng.random((1_000_000,5)).astype(np.float32); q5=rng.random((200_000,5)).astype(np.float32)
t5=pt.KdTree(p5, pt.Metric.L2Squared, 10, device=0)
print("f32 5-D 200k:", {k: timeit(t5,torch.from_numpy(q5).cuda(),k) for k in (16,32,33,40,64)}, flush=True)
p3=rng.random((1_000_000,3)).astype(np.float32); q3=rng.random((200_000,3)).astype(np.float32)
ts=pt.KdTree(p3, pt.Metric.SE2Squared, 10, device=0)
print("f32 SE2 200k:", {k: timeit(ts,torch.from_numpy(q3).cuda(),k) for k in (16,32,33,40)}, flush=True)
