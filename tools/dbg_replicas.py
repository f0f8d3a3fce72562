import os, sys, faulthandler
faulthandler.enable()
faulthandler.dump_traceback_later(90, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
for n in (3, 8):
    os.environ["PTK_MULTI_ALLOW_REPLICAS"] = "1"
    pts, q = ds.lidar_cloud(120_000, 1), ds.lidar_cloud(50_021, 2, pose=(3.0, 1.5))
    ref = oracle.Oracle(pts, 10, "port")
    ref.set_threads(ref.max_threads())
    print(n, "create", flush=True)
    multi = pt.MultiKdTree(pts, 10, devices=[0] * n)
    for name, batch in (("full", q), ("five", q[:5])):
        want1, want8 = ref.search_knn(batch, 1)[:, 0], ref.search_knn(batch, 8)
        print(n, name, "host k1", flush=True); assert multi.search_knn(batch, 1).tobytes() == want1.tobytes()
        print(n, name, "host k8", flush=True); assert multi.search_knn(batch, 8).tobytes() == want8.tobytes()
        dq = torch.from_numpy(batch).to("cuda:0")
        for k, want in ((1, want1), (8, want8)):
            print(n, name, "device k", k, flush=True)
            rows = multi.search_knn(dq, k).numpy(); torch.cuda.synchronize()
            assert rows.reshape(want.shape).tobytes() == want.tobytes()
        print(n, name, "radius", flush=True); got = multi.search_radius(batch, 0.02)
        off, flat = ref.search_radius(batch, 0.02)
        assert np.array_equal(got.offsets, off) and got.flat.tobytes() == flat.tobytes()
    del os.environ["PTK_MULTI_ALLOW_REPLICAS"]
    try:
        pt.MultiKdTree(pts, 10, devices=[0, 0])
    except Exception as e:
        print(n, "refused:", str(e)[:60], flush=True)
    print(n, "destroy", flush=True)
    del multi
    print(n, "done", flush=True)
