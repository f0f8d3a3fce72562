#!/usr/bin/env python3
"""The family the soak of r05 failed on (notes r05 item 24): points on a line in 2-D / 3-D, tiny leaves, queries off
the line -- thousands of points at nearly the same distance, trees a hundred levels deep, box distances that drift by
rounding.  k > 1 with the cap on every batch, against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PTK_TEST_KNOBS", "knn_cap_min_nq=1")
import oracle
import pico_tree_amd as pt

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
JITTER = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
bad = 0
for case in range(cases):
    rng = np.random.default_rng([seed, case])
    dim = int(rng.choice([2, 3])); n = int(rng.choice([3000, 20000, 60000])); nq = int(rng.choice([64, 300]))
    leaf = int(rng.choice([1, 2, 5])); scale = float(rng.choice([1.0, 37.5, 1e3]))
    t = rng.random((n, 1)); pts = ((t * rng.random((1, dim)) + 0.25) * scale).astype(np.float32)
    if JITTER:  # (no two distances equal: the long searches then end in the FIRST sweep's certificate)
        pts = (pts + rng.normal(0, JITTER, pts.shape) * scale).astype(np.float32)
    if rng.random() < 0.5:
        tq = rng.random((nq, 1)); q = ((tq * rng.random((1, dim)) + 0.25) * scale).astype(np.float32)
    else:
        q = (pts[rng.integers(0, n, nq)] + rng.normal(0, 1e-3, (nq, dim)) * scale).astype(np.float32)
    try:
        tree = pt.KdTree(pts, pt.Metric.L2Squared, leaf, device=0)
    except pt.PtkError:
        continue
    ref = oracle.Oracle(pts, leaf, "port")
    for k in {int(rng.choice([2, 5, 16])), int(rng.choice([24, 32, 33, 48, 56]))}:
        k = min(k, n)
        if tree.search_knn(q, k).tobytes() != ref.search_knn(q, k).tobytes():
            bad += 1
            print(f"FAIL lines seed {seed} case {case}: dim {dim} n {n} nq {nq} leaf {leaf} scale {scale} k {k}", flush=True)
    tree.close()
print(f"fuzz_lines: {cases} cases, seed {seed}, jitter {JITTER}, {bad} failing")
