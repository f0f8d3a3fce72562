"""kd_forest (BASELINE config 5): the product's forest search against

* its CPU restatement (oracle.ForestOracle) -- bit for bit, since both use the same reflection
  vectors, the same de-duplicated k-list and original-space distances;
* the exact kd_tree -- recall, the quality measure the reference itself reports
  (/root/reference/examples/kd_forest/kd_forest.cpp:113-123);
* the compiled reference kd_forest -- recall of both against the same exact answers (the
  reference draws its reflections from std::random_device, so only statistics compare).

The CPU tier runs the real kernel source under the emulator; the `gpu` tier runs libptk.
"""

from __future__ import annotations

import numpy as np
import pytest

import oracle
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
from tests.emu import emulated_forest_knn


def _recall(found, exact):
    """Mean fraction of the exact k nearest that appear in the found rows."""
    hits = [len(set(f["index"].tolist()) & set(e["index"].tolist())) / len(e) for f, e in zip(found, exact)]
    return float(np.mean(hits))


def _clouds(n, nq, dim):
    return ds.sift_like_cloud(n, dim, seed=1, centres=60), ds.sift_like_cloud(nq, dim, seed=2, centres=60)


@pytest.mark.parametrize("dim,n,nq,leaf,trees,k,leaves", [(16, 4_000, 96, 8, 3, 5, 6), (32, 3_000, 64, 16, 4, 10, 8),
                                                          (128, 2_500, 48, 32, 2, 1, 4), (5, 3_000, 64, 4, 5, 64, 3)])
def test_emulated_forest_kernel_equals_oracle(dim, n, nq, leaf, trees, k, leaves):
    pts, q = _clouds(n, nq, dim)
    got, rot, dropped = emulated_forest_knn(pts, leaf, trees, 7, q, k, leaves)
    assert dropped == 0
    assert np.allclose(np.linalg.norm(rot, axis=1), 1.0, atol=1e-6)
    want = oracle.ForestOracle(pts, leaf, rot).search_knn(q, k, leaves)
    assert got.tobytes() == want.tobytes()
    # rows ascending, indices distinct
    for row in got:
        real = row[row["index"] >= 0]
        assert np.all(np.diff(real["distance"]) >= 0) and len(set(real["index"].tolist())) == len(real)


def test_forest_oracle_recall_matches_the_compiled_reference():
    """Same data, same (trees, leaf, leaves): recall@1 of the restated forest and of the actual
    reference forest against the exact kd_tree agree within sampling noise."""
    if not oracle.have_reference_forest():
        pytest.skip("oracle/_ref/libptk_ref_forest.so is only built where /root/reference exists")
    pts, q = _clouds(20_000, 400, 32)
    exact = oracle.Oracle(pts, 10, "port").search_knn(q, 1)
    trees, leaf, leaves = 4, 16, 8
    rot = np.stack([emulated_forest_knn(pts[:64], 4, trees, 3, q[:1], 1, 1)[1][i] for i in range(trees)])
    ours = oracle.ForestOracle(pts, leaf, rot).search_knn(q, 1, leaves)
    ref = oracle.ReferenceForest(pts, leaf, trees).search_knn(q, 1, leaves)
    r_ours, r_ref = _recall(ours, exact), _recall(ref, exact)
    assert r_ours > 0.5 and abs(r_ours - r_ref) < 0.08, (r_ours, r_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,n,nq,leaf,trees,k,leaves", [(128, 30_000, 700, 32, 8, 10, 64), (128, 30_000, 700, 32, 8, 1, 64),
                                                          (16, 50_000, 2_000, 8, 4, 5, 10), (64, 20_000, 500, 16, 3, 64, 16),
                                                          (7, 20_000, 1_000, 4, 6, 3, 5)])
def test_forest_bit_exact_against_the_oracle(gpu, dim, n, nq, leaf, trees, k, leaves):
    pts, q = _clouds(n, nq, dim)
    forest = pt.KdForest(pts, leaf, trees, seed=11, device=gpu)
    got = forest.search_knn(q, k, leaves)
    want = oracle.ForestOracle(pts, leaf, forest.rotations).search_knn(q, k, leaves)
    assert got.tobytes() == want.tobytes()
    if k == 1:
        assert forest.search_nn(q, leaves).tobytes() == want[:, 0].tobytes()


@pytest.mark.gpu
def test_forest_recall_and_device_buffers(gpu):
    import torch

    pts, q = _clouds(60_000, 1_000, 128)
    forest = pt.KdForest(pts, 32, 8, seed=5, device=gpu)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    exact = tree.search_knn(q, 10)
    got = forest.search_knn(q, 10, 64)
    assert _recall(got[:, :1], exact[:, :1]) > 0.9
    assert _recall(got, exact) > 0.8
    # more leaves never hurt; one leaf per tree is much worse
    assert _recall(forest.search_knn(q, 10, 2), exact) < _recall(got, exact)
    dq = torch.from_numpy(q).to(f"cuda:{gpu}")
    assert forest.search_knn(dq, 10, 64).numpy().tobytes() == got.tobytes()
    # distances are true squared distances in the original space
    i = got["index"][:, 0]
    d = ((q - pts[i]) ** 2).sum(1)
    assert np.allclose(d, got["distance"][:, 0], rtol=1e-5)


@pytest.mark.gpu
def test_forest_config5_at_full_size(gpu):
    """BASELINE configs[4] at its own size: 1 M x 128 points (SIFT-like synthetic), 10 k queries,
    8 trees, leaf 32, 64 leaves per tree, k = 10 (the reference's SIFT setting,
    examples/kd_forest/kd_forest.cpp:113-123).  Recall against the EXACT neighbours (brute force on
    the GPU, test-side torch), no queue entry dropped, and bit-equality with the restated forest
    on a 500-query sample."""
    import torch

    n, nq, dim, k = 1_000_000, 10_000, 128, 10
    pts, q = ds.sift_like_cloud(n, dim, seed=1), ds.sift_like_cloud(nq, dim, seed=2)
    forest = pt.KdForest(pts, 32, 8, seed=1, device=gpu)
    got = forest.search_knn(q, k, 64)
    assert forest.dropped == 0
    dev = torch.device("cuda", gpu)
    dp = torch.from_numpy(pts).to(dev)
    pn = (dp.double() ** 2).sum(1)
    exact = np.empty((nq, k), dtype=np.int64)
    for lo in range(0, nq, 500):  # float64 distances: the values are integers below 2^24, so this is exact
        dq = torch.from_numpy(q[lo:lo + 500]).to(dev).double()
        d2 = (dq ** 2).sum(1)[:, None] - 2.0 * (dq @ dp.double().T) + pn[None, :]
        exact[lo:lo + 500] = torch.topk(d2, k, dim=1, largest=False).indices.cpu().numpy()
        del d2
    r1 = float(np.mean(got["index"][:, 0] == exact[:, 0]))
    r10 = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / k for a, b in zip(got["index"], exact)]))
    assert r1 >= 0.93 and r10 >= 0.90, (r1, r10)
    sample = np.arange(0, nq, nq // 500)[:500]
    want = oracle.ForestOracle(pts, 32, forest.rotations).search_knn(q[sample], k, 64)
    assert got[sample].tobytes() == want.tobytes()


@pytest.mark.gpu
def test_cpp_kd_forest_header(gpu, tmp_path):
    """include/pico_understory/kd_forest.hpp (the reference's class shape) through the C ABI:
    per-query search_nn, batched search_nn / search_knn, all equal to the Python binding and
    therefore to the oracle."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pts, q = _clouds(20_000, 300, 16)
    pts.tofile(tmp_path / "points.bin")
    q.tofile(tmp_path / "queries.bin")
    exe = str(tmp_path / "forest_main")
    libdir = os.path.join(root, "pico_tree_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "forest_main.cpp"), "-o", exe,
                           "-L" + libdir, "-lptk", "-Wl,-rpath," + libdir])
    env = dict(os.environ, HIP_VISIBLE_DEVICES=str(gpu)) if gpu else None
    subprocess.check_call([exe, str(tmp_path)], env=env)
    forest = pt.KdForest(pts, 8, 4, seed=11, device=gpu)
    want1 = forest.search_nn(q, 10)
    want5 = forest.search_knn(q, 5, 10)
    one = np.fromfile(tmp_path / "f_one.bin", dtype=pt.NEIGHBOR)
    batch = np.fromfile(tmp_path / "f_batch.bin", dtype=pt.NEIGHBOR)
    knn = np.fromfile(tmp_path / "f_knn.bin", dtype=pt.NEIGHBOR).reshape(len(q), 5)
    assert one.tobytes() == want1.tobytes() and batch.tobytes() == want1.tobytes()
    assert knn.tobytes() == want5.tobytes()
    assert want5.tobytes() == oracle.ForestOracle(pts, 8, forest.rotations).search_knn(q, 5, 10).tobytes()
