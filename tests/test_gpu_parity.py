"""GPU parity tests: the HIP path, called through the C ABI (ctypes -> libptk.so),
against the CPU oracle on the same seeded inputs.  Bar: bit-exact indices AND
bit-exact float32 distances (the oracle is built without FMA contraction).

Index parity is pinned by the oracle only: the reference's own tests compare
distances but deliberately not indices
(/root/reference/test/pico_tree/common.hpp:195-196); the oracle in turn is pinned
against the compiled reference (tests/test_oracle.py).
"""

from __future__ import annotations

import numpy as np
import pytest

import oracle
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

pytestmark = pytest.mark.gpu


def _clouds(name, n, nq):
    if name == "uniform":
        return ds.uniform_cloud(n, 3, 11), ds.uniform_cloud(nq, 3, 12)
    if name == "lidar":
        return ds.lidar_cloud(n, 1), ds.lidar_cloud(nq, 2, pose=(3.0, 1.5))
    if name == "ties":  # coordinates on a coarse lattice: many equal distances, duplicates
        return (np.round(ds.uniform_cloud(n, 3, 5) * 8) / 8).astype(np.float32), \
               (np.round(ds.uniform_cloud(nq, 3, 6) * 16) / 16).astype(np.float32)
    if name == "self":  # queries are tree points: zero distances, exact plane hits
        p = ds.uniform_cloud(n, 3, 21)
        return p, p[:nq].copy()
    raise ValueError(name)


@pytest.fixture(scope="module")
def trees(gpu):
    cache = {}

    def get(name, n=60_000, nq=20_000, leaf=10):
        key = (name, n, nq, leaf)
        if key not in cache:
            pts, q = _clouds(name, n, nq)
            cache[key] = (pt.KdTree(pts, pt.Metric.L2Squared, leaf, device=gpu),
                          oracle.Oracle(pts, leaf, "port"), pts, q)
        return cache[key]

    return get


@pytest.mark.parametrize("cloud", ["uniform", "lidar", "ties", "self"])
@pytest.mark.parametrize("k", [1, 4, 16, 40])
def test_knn_bit_exact(trees, cloud, k):
    tree, ref, _, q = trees(cloud)
    got = tree.search_knn(q, k)
    want = ref.search_knn(q, k)
    if k == 1:
        want = want[:, 0]
    assert got.shape == want.shape
    assert np.array_equal(got["index"], want["index"])
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("cloud", ["uniform", "ties"])
def test_reorder_does_not_change_results(trees, cloud):
    tree, _, _, q = trees(cloud)
    tree.set_reorder(pt.REORDER_OFF)
    a1, a8 = tree.search_knn(q, 1), tree.search_knn(q, 8)
    tree.set_reorder(pt.REORDER_ON)
    b1, b8 = tree.search_knn(q, 1), tree.search_knn(q, 8)
    tree.set_reorder(pt.REORDER_AUTO)
    assert a1.tobytes() == b1.tobytes()
    assert a8.tobytes() == b8.tobytes()


@pytest.mark.parametrize("cloud,radius", [("uniform", 0.0015), ("lidar", 1.0), ("ties", 0.03)])
def test_radius_bit_exact_traversal_order(trees, cloud, radius):
    tree, ref, _, q = trees(cloud)
    got = tree.search_radius(q, radius)
    off, flat = ref.search_radius(q, radius)
    assert np.array_equal(got.offsets, off)
    assert got.flat.tobytes() == flat.tobytes()
    assert off[-1] > 0


@pytest.mark.parametrize("leaf", [1, 33, 100])
def test_radius_rows_from_leaf_lists_with_other_leaf_sizes(gpu, leaf):
    """The radius search of a 3-D tree lists, per query, the leaves with hits and the mask of the hits (32 points per
    list entry: a larger leaf is listed in pieces) and makes the rows by replaying the lists
    (ptk_kernels_lists.hpp).  Leaves of one point, of 33 and of 100; long rows (queries on top of dense spots), empty
    rows, a partly empty last wavefront; exact, approximate and sorted -- rows equal the oracle's."""
    import torch

    pts, q = ds.lidar_cloud(150_000, seed=3), ds.lidar_cloud(20_000 - 37, seed=4, pose=(3.0, 1.5))
    q[:500] = pts[:500]
    tree = pt.KdTree(pts, pt.Metric.L2Squared, leaf, device=gpu)
    ref = oracle.Oracle(pts, leaf, "port")
    ref.set_threads(ref.max_threads())
    dq = torch.from_numpy(q).to(f"cuda:{gpu}")
    most = 0
    for radius, e in ((4.0, 1.0), (0.05, 1.0), (1.5, 1.4)):
        want_off, want = ref.search_radius(q, radius, e=None if e == 1.0 else e)
        off, raw = tree.search_radius_device(dq, radius, e)
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
        assert raw.cpu().numpy().tobytes() == want.tobytes()
        most = max(most, int(want_off[-1]))
    assert most > 5 * len(q), most
    got = tree.search_radius(q, 4.0, sort=True)
    _, want = ref.search_radius(q, 4.0, sort=True)
    assert np.array_equal(got.flat["distance"], want["distance"])


def test_radius_lists_longer_than_a_wavefront_may_hold(gpu):
    """A query whose row draws on more than 1 024 leaves (kListMaxChunks chunks of 16 entries per lane) cannot be
    listed: its wavefront is marked and its queries take the ordinary fill traversal, the wavefronts beside it the
    replay -- rows equal the oracle's either way."""
    pts = ds.uniform_cloud(100_000, 3, 21)
    q = np.ascontiguousarray(np.concatenate([ds.uniform_cloud(640, 3, 22) * 0.2 + 0.4, ds.uniform_cloud(3_000, 3, 23)]))
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 2, device=gpu)
    ref = oracle.Oracle(pts, 2, "port")
    ref.set_threads(ref.max_threads())
    for radius in (0.04, 0.0005):  # ~3 000 hits in ~1 500 leaves per central query / a handful
        want_off, want = ref.search_radius(q, radius)
        got = tree.search_radius(q, radius)
        assert np.array_equal(got.offsets, want_off) and got.flat.tobytes() == want.tobytes()
    assert int(np.diff(ref.search_radius(q[:640], 0.04)[0]).min()) > 2 * 1024


def test_radius_strict_and_sorted(trees):
    tree, ref, pts, q = trees("ties")
    radius = 0.03
    got = tree.search_radius(q, radius, sort=True)
    off, flat = ref.search_radius(q, radius, sort=True)
    assert np.array_equal(got.offsets, off)
    # std::sort is unstable: rows agree as sorted distance sequences and as index sets.
    assert np.array_equal(got.flat["distance"], flat["distance"])
    for i in range(0, len(q), 97):
        a, b = got[i], flat[int(off[i]):int(off[i + 1])]
        assert sorted(a["index"].tolist()) == sorted(b["index"].tolist())
        assert np.all(np.diff(a["distance"]) >= 0)
        assert np.all(a["distance"] < np.float32(radius))  # strict (search_visitor.hpp:141)


@pytest.mark.parametrize("e", [1.21, 2.0])
def test_approximate_searches(trees, e):
    tree, ref, _, q = trees("lidar")
    assert tree.search_knn(q, 8, e).tobytes() == ref.search_knn(q, 8, e=e).tobytes()
    assert tree.search_knn(q, 1, e).tobytes() == ref.search_knn(q, 1, e=e)[:, 0].tobytes()
    got = tree.search_radius(q, 2.0, e)
    off, flat = ref.search_radius(q, 2.0, e=e)
    assert np.array_equal(got.offsets, off) and got.flat.tobytes() == flat.tobytes()


@pytest.mark.parametrize("dim", [1, 2])
def test_low_dimensions(gpu, dim):
    pts, q = ds.uniform_cloud(30_000, dim, 3), ds.uniform_cloud(10_000, dim, 4)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 7, device=gpu)
    ref = oracle.Oracle(pts, 7, "port")
    assert tree.search_knn(q, 5).tobytes() == ref.search_knn(q, 5).tobytes()
    got = tree.search_radius(q, 1e-4 if dim == 2 else 1e-6)
    off, flat = ref.search_radius(q, 1e-4 if dim == 2 else 1e-6)
    assert np.array_equal(got.offsets, off) and got.flat.tobytes() == flat.tobytes()


@pytest.mark.parametrize("dim,n,nq,radius", [(4, 30_000, 8_000, 0.01), (5, 30_000, 8_000, 0.03), (8, 20_000, 4_000, 0.15),
                                             (16, 20_000, 2_000, 0.9), (64, 5_000, 500, 8.0),
                                             (128, 4_000, 300, 18.0)])
def test_any_dimension(gpu, dim, n, nq, radius):
    """dim > 3 runs the any-dimension kernels (query and offsets staged in LDS)."""
    pts, q = ds.uniform_cloud(n, dim, 3), ds.uniform_cloud(nq, dim, 4)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 9, device=gpu)
    ref = oracle.Oracle(pts, 9, "port")
    assert tree.search_knn(q, 1).tobytes() == ref.search_knn(q, 1)[:, 0].tobytes()
    for k in (6, 40):
        assert tree.search_knn(q, k).tobytes() == ref.search_knn(q, k).tobytes()
    assert tree.search_knn(q, 5, 1.3).tobytes() == ref.search_knn(q, 5, e=1.3).tobytes()
    got = tree.search_radius(q, radius)
    off, flat = ref.search_radius(q, radius)
    assert np.array_equal(got.offsets, off) and got.flat.tobytes() == flat.tobytes()
    assert off[-1] > 0


@pytest.mark.parametrize("name", ["g_small_3d", "g_small_2d", "g_small_5d", "g_ties_3d"])
def test_golden_vectors_on_the_device(gpu, name):
    """The committed fixtures (outputs of the compiled reference) through the HIP path."""
    from tests.test_oracle import check_against_golden, load

    g = load(name)

    class Adaptor:
        def __init__(self):
            self.tree = pt.KdTree(np.ascontiguousarray(g["points"]), pt.Metric.L2Squared, int(g["max_leaf_size"]),
                                  device=gpu)

        def search_knn(self, q, k, e=None):
            r = self.tree.search_knn(np.ascontiguousarray(q), k, *(() if e is None else (e,)))
            return r[:, None] if k == 1 else r

        def search_radius(self, q, radius, sort=False, e=None):
            r = self.tree.search_radius(np.ascontiguousarray(q), radius, *(() if e is None else (e,)), sort=sort)
            return r.offsets, r.flat

    check_against_golden(Adaptor(), g)


@pytest.mark.parametrize("cloud,dim,half", [("uniform", 3, 0.03), ("lidar", 3, 1.5), ("ties", 3, 0.125), ("u2", 2, 0.02),
                                            ("u1", 1, 0.001), ("u4", 4, 0.15), ("u7", 7, 0.35), ("u24", 24, 0.6)])
def test_box_search_traversal_order(gpu, cloud, dim, half):
    """search_box on the device: rows equal the reference's traversal-order index lists
    (kd_tree_search.hpp:238-381), including wholesale-reported subtrees and closed bounds."""
    if cloud in ("uniform", "lidar", "ties"):
        pts, q = _clouds(cloud, 60_000, 6_000)
    else:
        pts, q = ds.uniform_cloud(40_000, dim, 3), ds.uniform_cloud(4_000, dim, 4)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    ref = oracle.Oracle(pts, 10, "port")
    rng = np.random.default_rng(5)
    h = (rng.uniform(0.2, 1.0, size=q.shape) * half).astype(np.float32)
    mins, maxs = q - h, q + h
    mins[::50] = pts.min(0) - 1  # a few boxes swallow the whole cloud or large subtrees
    maxs[::50] = pts.max(0) + 1
    maxs[1::50] = mins[1::50]    # and a few are degenerate (min == max)
    boxes = np.empty((2 * len(q), dim), dtype=np.float32)
    boxes[0::2], boxes[1::2] = mins, maxs
    got = tree.search_box(boxes)
    off, flat = ref.search_box(mins, maxs)
    assert np.array_equal(got.offsets, off)
    assert np.array_equal(got.flat, flat)
    assert off[-1] > len(pts)


def test_deep_tree_uses_scratch_overflow(gpu):
    """Heavy duplication makes the sliding-midpoint tree > 100 levels deep."""
    pts = (np.round(ds.uniform_cloud(40_000, 3, 9) * 4) / 4).astype(np.float32)
    q = ds.uniform_cloud(10_000, 3, 10)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 4, device=gpu)
    assert tree.info()["max_depth"] > 60
    ref = oracle.Oracle(pts, 4, "port")
    assert tree.search_knn(q, 1).tobytes() == ref.search_knn(q, 1)[:, 0].tobytes()
    assert tree.search_knn(q, 10).tobytes() == ref.search_knn(q, 10).tobytes()


@pytest.mark.parametrize("n,leaf", [(1, 1), (7, 10), (64, 1), (1000, 1000)])
def test_tiny_and_degenerate_trees(gpu, n, leaf):
    pts, q = ds.uniform_cloud(n, 3, 31), ds.uniform_cloud(500, 3, 32)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, leaf, device=gpu)
    ref = oracle.Oracle(pts, leaf, "port")
    for k in {1, min(3, n), n if n <= 64 else 50}:
        got, want = tree.search_knn(q, k), ref.search_knn(q, k)
        assert got.tobytes() == (want[:, 0] if k == 1 else want).tobytes()


def test_device_buffers_and_streams(trees, gpu):
    import torch

    tree, ref, _, q = trees("uniform")
    dq = torch.from_numpy(q).to(f"cuda:{gpu}")
    side = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(side):
        res = tree.search_knn(dq, 4)
    side.synchronize()
    assert res.numpy().tobytes() == ref.search_knn(q, 4).tobytes()
    off, raw = tree.search_radius_device(dq, 0.0015)
    torch.cuda.synchronize()
    o2, flat = ref.search_radius(q, 0.0015)
    assert np.array_equal(off.cpu().numpy().astype(np.uint64), o2)
    assert raw.cpu().numpy().tobytes() == flat.tobytes()


def test_large_batches_go_through_in_pieces(trees, monkeypatch):
    """ptk_search_knn_device cuts batches above PTK_MAX_BATCH (default 2^25) into pieces."""
    tree, ref, _, q = trees("lidar")
    monkeypatch.setenv("PTK_MAX_BATCH", "3001")
    assert tree.search_knn(q, 1).tobytes() == ref.search_knn(q, 1)[:, 0].tobytes()
    assert tree.search_knn(q, 6).tobytes() == ref.search_knn(q, 6).tobytes()


def test_empty_batch_and_errors(trees):
    tree, _, pts, q = trees("uniform")
    assert tree.search_knn(q[:0], 3).shape == (0, 3)
    assert len(tree.search_radius(q[:0], 1.0)) == 0
    with pytest.raises(pt.PtkError):
        tree.search_knn(q, 0)
    with pytest.raises(ValueError):  # (k > n_points is served as the reference serves it: test_k_larger_than_the_tree)
        tree.search_knn(q.astype(np.float64), 1)
    with pytest.raises(ValueError):
        tree.search_knn(q[:, :2].copy(), 1)
    with pytest.raises(ValueError):
        tree.search_knn(q[::2], 1)  # non-contiguous


def test_subnormal_and_huge_coordinates(gpu):
    """float32 subnormals must not be flushed and 1e18-scale values must not overflow early."""
    base = ds.uniform_cloud(20_000, 3, 41)
    for scale in (np.float32(1e-38), np.float32(1e-20), np.float32(1e15)):
        pts = (base * scale).astype(np.float32)
        q = (ds.uniform_cloud(5_000, 3, 42) * scale).astype(np.float32)
        tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
        ref = oracle.Oracle(pts, 10, "port")
        assert tree.search_knn(q, 3).tobytes() == ref.search_knn(q, 3).tobytes()


@pytest.mark.parametrize("metric", ["L1", "LPInf", "LNInf"])
@pytest.mark.parametrize("cloud,dim", [("uniform", 3), ("lidar", 3), ("ties", 3), ("u2", 2), ("u6", 6), ("u32", 32)])
def test_other_metrics_bit_exact(gpu, cloud, dim, metric):
    """metric_l1 / metric_lpinf (metric.hpp:78-152) through ptk_tree_set_metric: same tree, the
    generic kernels with the metric swapped in; the oracle under the same metric is pinned against
    kd_tree<space, metric_l1|metric_lpinf> of the compiled reference (tests/test_oracle.py)."""
    if dim == 3:
        pts, q = _clouds(cloud, 50_000, 12_000)
    else:
        n, nq = (40_000, 8_000) if dim <= 6 else (8_000, 1_000)
        pts, q = ds.uniform_cloud(n, dim, 71), ds.uniform_cloud(nq, dim, 72)
    tree = pt.KdTree(pts, pt.Metric[metric], 10, device=gpu)
    ref = oracle.Oracle(pts, 10, "port", metric)
    assert tree.search_knn(q, 1).tobytes() == ref.search_knn(q, 1)[:, 0].tobytes()
    for k in (5, 16, 40):
        assert tree.search_knn(q, k).tobytes() == ref.search_knn(q, k).tobytes()
    assert tree.search_knn(q, 6, 1.25).tobytes() == ref.search_knn(q, 6, e=1.25).tobytes()
    scale = float(np.ptp(pts, axis=0).max())
    radius = scale * {("L1", 2): 0.01, ("L1", 3): 0.02, ("L1", 6): 0.45, ("L1", 32): 7.0,
                      ("LPInf", 2): 0.01, ("LPInf", 3): 0.02, ("LPInf", 6): 0.15, ("LPInf", 32): 0.5,
                      ("LNInf", 2): 0.0002, ("LNInf", 3): 0.0002, ("LNInf", 6): 0.0002, ("LNInf", 32): 0.0002}[(metric, dim)]
    got = tree.search_radius(q, radius)
    off, flat = ref.search_radius(q, radius)
    assert np.array_equal(got.offsets, off) and got.flat.tobytes() == flat.tobytes()
    assert off[-1] > 0
    got = tree.search_radius(q, radius, 1.5, sort=True)
    off, flat = ref.search_radius(q, radius, e=1.5, sort=True)
    assert np.array_equal(got.offsets, off) and np.array_equal(got.flat["distance"], flat["distance"])
    # the same handle keeps answering box queries (metric-free)
    if dim == 3:
        half = np.float32(0.02 * scale)
        boxes = np.empty((4000, 3), dtype=np.float32)
        boxes[0::2], boxes[1::2] = q[:2000] - half, q[:2000] + half
        b = tree.search_box(boxes)
        boff, bflat = ref.search_box(q[:2000] - half, q[:2000] + half)
        assert np.array_equal(b.offsets, boff) and np.array_equal(b.flat, bflat)


def test_metric_round_trips_through_saved_file(gpu, tmp_path):
    pts, q = ds.uniform_cloud(20_000, 3, 81), ds.uniform_cloud(3_000, 3, 82)
    tree = pt.KdTree(pts, pt.Metric.L1, 8, device=gpu)
    path = str(tmp_path / "l1.pkd")
    pt.save_kd_tree(tree, path)
    again = pt.load_kd_tree(pts, path, device=gpu)
    assert "metric=L1" in repr(again)
    assert again.search_knn(q, 4).tobytes() == tree.search_knn(q, 4).tobytes()
    assert again.search_knn(q, 4).tobytes() == oracle.Oracle(pts, 8, "port", "L1").search_knn(q, 4).tobytes()


@pytest.mark.parametrize("env", [{}, {"radius_capture_chunks": 0}, {"radius_capture_chunks": 1},
                                 {"PTK_RADIUS_CAPTURE_MB": "0"}, {"PTK_RADIUS_CAPTURE_MB": "2"}],
                         ids=["default", "static-chunk-only", "pool-runs-dry", "capture-off", "budget-too-small"])
@pytest.mark.parametrize("cloud,radius", [("lidar", 1.0), ("ties", 0.03)])
def test_radius_rows_captured_in_the_count_pass(trees, monkeypatch, env, cloud, radius):
    """The count pass captures the rows and the fill pass copies them (RadiusCapture in
    ptk_kernels.hpp); rows that did not fit are searched again.  Same bytes in every regime."""
    import torch
    for name, value in env.items():  # (an environment switch of the library, or one of its test hooks)
        if name.startswith("PTK_"):
            monkeypatch.setenv(name, value)
        else:
            pt.set_test_knobs(**{name: value})
    tree, ref, _, q = trees(cloud)
    want_off, want = ref.search_radius(q, radius)
    got = tree.search_radius(q, radius)
    assert np.array_equal(got.offsets, want_off) and got.flat.tobytes() == want.tobytes()
    dq = torch.from_numpy(q).cuda()
    off, raw = tree.search_radius_device(dq, radius)
    assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
    assert raw.cpu().numpy().tobytes() == want.tobytes()
    want_off, want = ref.search_radius(q, radius, e=1.5, sort=True)
    off, raw = tree.search_radius_device(dq, radius, 1.5, sort=True)
    assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
    assert np.array_equal(raw.cpu().numpy()[:, 1].view(np.float32), want["distance"])


def test_tree_deeper_than_the_private_stack_classes(gpu, monkeypatch):
    """1 500 coincident points peel one level each (sliding midpoint): depth > 1 031 is past the
    private spill classes, so the record stacks of these searches spill to HBM (the
    reference builds and searches such clouds -- LiDAR scans carry thousands of (0, 0, 0) returns)."""
    pts = np.concatenate([ds.uniform_cloud(60_000, 3, 31) - np.float32(0.5), np.zeros((1_500, 3), np.float32)])
    q = np.concatenate([ds.uniform_cloud(3_000, 3, 32) - np.float32(0.5), np.zeros((3, 3), np.float32),
                        np.full((2, 3), 1e-3, np.float32)])
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    assert tree.info()["max_depth"] > 1_040
    ref = oracle.Oracle(pts, 10, "port")
    monkeypatch.setenv("PTK_DEEP_SPILL_MB", "16")  # several launches per batch
    for k in (1, 5, 40):
        want = ref.search_knn(q, k)
        assert tree.search_knn(q, k).tobytes() == (want[:, 0] if k == 1 else want).tobytes()
    got = tree.search_radius(q, 0.002)
    off, flat = ref.search_radius(q, 0.002)
    assert np.array_equal(got.offsets, off) and got.flat.tobytes() == flat.tobytes()
    boxes = np.empty((2 * len(q), 3), dtype=np.float32)
    boxes[0::2], boxes[1::2] = q - np.float32(0.02), q + np.float32(0.02)
    gb = tree.search_box(boxes)
    boff, bflat = ref.search_box(boxes[0::2].copy(), boxes[1::2].copy())
    assert np.array_equal(gb.offsets, boff) and np.array_equal(gb.flat, bflat)


def test_k_larger_than_the_tree(gpu):
    """k > n_points: the reference's iterator-range search_knn (and its Python binding) fills the n
    neighbours and leaves the FLT_MAX sentinel in the last slot (search_visitor.hpp:95-110)."""
    pts, q = ds.uniform_cloud(7, 3, 41), ds.uniform_cloud(50, 3, 42)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 3, device=gpu)
    ref = oracle.Oracle(pts, 3, "port")
    got = tree.search_knn(q, 12)
    want = ref.search_knn(q, 7)
    assert got.shape == (50, 12)
    assert got[:, :7].tobytes() == want.tobytes()
    assert np.all(got["distance"][:, 11] == np.float32(3.402823466e+38))


def test_host_radius_passes_do_not_reuse_a_stale_capture(trees):
    """count(A), count(B), fill(A) through the HOST forms: every call uploads into a fresh device
    buffer, which the allocator hands out at the same address again -- the capture of count(B) must
    not serve fill(A)."""
    tree, ref, _, q = trees("uniform")
    lib = pt._load()
    import ctypes
    radius = 0.0015
    a, b = np.ascontiguousarray(q[:9000]), np.ascontiguousarray(q[9000:18000])

    def count(x):
        c = np.zeros(len(x), dtype=np.uint64)
        assert lib.ptk_search_radius_count(tree._h, x.ctypes.data, len(x), ctypes.c_float(radius), ctypes.c_float(1.0),
                                           c.ctypes.data) == 0
        off = np.zeros(len(x) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(c)
        return off

    def fill(x, off):
        out = np.zeros(max(int(off[-1]), 1), dtype=pt.NEIGHBOR)
        assert lib.ptk_search_radius_fill(tree._h, x.ctypes.data, len(x), ctypes.c_float(radius), ctypes.c_float(1.0),
                                          off.ctypes.data, out.ctypes.data, 0) == 0
        return out[:int(off[-1])]

    off_a = count(a)
    off_b = count(b)
    want_off, want = ref.search_radius(a, radius)
    assert np.array_equal(off_a, want_off)
    assert fill(a, off_a).tobytes() == want.tobytes()
    want_off, want = ref.search_radius(b, radius)
    assert np.array_equal(off_b, want_off) and fill(b, off_b).tobytes() == want.tobytes()


def test_radius_fill_that_does_not_match_the_last_count(trees):
    """count(A), count(B), fill(A): the capture belongs to B, so A's fill must search again."""
    import ctypes
    import torch
    tree, ref, _, q = trees("uniform")
    lib = pt._load()
    radius = 0.0015
    a, b = torch.from_numpy(q[:9000]).cuda(), torch.from_numpy(q[9000:18000].copy()).cuda()
    stream = torch.cuda.current_stream().cuda_stream

    def count(t):
        c = torch.zeros(len(t) + 1, dtype=torch.int64, device="cuda")
        assert lib.ptk_search_radius_count_device(tree._h, t.data_ptr(), len(t), ctypes.c_float(radius),
                                                  ctypes.c_float(1.0), c.data_ptr(), stream) == 0
        off = torch.zeros(len(t) + 1, dtype=torch.int64, device="cuda")
        off[1:] = torch.cumsum(c[:len(t)], 0)
        return off

    def fill(t, off):
        out = torch.empty((max(int(off[-1].item()), 1), 2), dtype=torch.int32, device="cuda")
        assert lib.ptk_search_radius_fill_device(tree._h, t.data_ptr(), len(t), ctypes.c_float(radius),
                                                 ctypes.c_float(1.0), off.data_ptr(), out.data_ptr(), 0, stream) == 0
        return out[:int(off[-1].item())].cpu().numpy()

    off_a, off_b = count(a), count(b)
    for t, off, rows in ((a, off_a, q[:9000]), (b, off_b, q[9000:18000]), (a, off_a, q[:9000])):
        want_off, want = ref.search_radius(rows, radius)
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off)
        assert fill(t, off).tobytes() == want.tobytes()


@pytest.mark.parametrize("cloud", ["L", "U"])
def test_full_size_results_hash_like_the_reference(gpu, cloud):
    """BASELINE config 2 / 3 at FULL size (7.73 M points, 7.2 M queries): SHA-256 of the device
    results against checksums of the COMPILED REFERENCE's results (tests/golden/hashes_full.json,
    written by tests/golden/make_full_hashes.py) -- knn = 1, knn = 16, radius rows in traversal
    order.  No oracle at test time."""
    import hashlib
    import json
    import os
    import torch
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hashes_full.json")
    if not os.path.exists(path):
        pytest.skip("hashes_full.json not generated")
    with open(path) as f:
        want = json.load(f)

    def sha(t):
        h = hashlib.sha256()
        a = t.contiguous().cpu().numpy().reshape(-1).view(np.uint8)
        for i in range(0, len(a), 1 << 28):
            h.update(a[i:i + (1 << 28)].tobytes())
        return h.hexdigest()

    pts, q = ds.config2_clouds(cloud)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    dq = torch.from_numpy(q).cuda()
    for k in (1, 16):
        out = torch.empty((len(q), k, 2), dtype=torch.int32, device="cuda")
        tree.search_knn(dq, k, out)
        e = want[f"config2_{cloud}_full_knn{k}"]
        assert int(out[:, :, 0].sum(dtype=torch.int64).item()) == e["index_sum"]
        assert sha(out[:, :, 0]) == e["index_sha256"]
        assert sha(out[:, :, 1]) == e["distance_bits_sha256"]
        del out
    off, raw = tree.search_radius_device(dq, 1.0)
    e = want[f"config3_{cloud}_full_radius1.0"]
    assert int(off[-1].item()) == e["hits"]
    assert sha(off) == e["offsets_sha256"]   # int64 here, uint64 there: same bytes
    assert sha(raw) == e["rows_sha256"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["float32", "float64"])
def test_randomised_differential_sweep(gpu, dtype):
    """A slice of tools/fuzz_parity.py (random dim 1-7, tree size, leaf size, cloud shape, scale,
    metric, k, e, radius, boxes, reorder mode) against the oracle; the full tool ran 2 550 cases
    on the MI355X with none failing (profiles/r01l_notes.txt).  float64: the same sweep through the
    double-precision entry points against the oracle's double build."""
    import importlib.util
    import os
    import torch
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    fuzz.DTYPE = dtype
    failing, ran = [], 0
    for case in range(80):
        rng = np.random.default_rng([11, case])
        try:
            desc, bad = fuzz.one_case(rng, pt, oracle, torch, case)
        except pt.PtkError as err:
            assert "too deep" in str(err) or "deeper than" in str(err) or "levels" in str(err), str(err)
            continue
        ran += 1
        if bad:
            failing.append((desc, bad))
    assert not failing, failing
    assert ran >= 60


@pytest.mark.parametrize("env", [{}, {"radius_capture_chunks": 0}, {"radius_capture_chunks": 1},
                                 {"PTK_RADIUS_CAPTURE_MB": "0"}], ids=["default", "static-chunk-only", "pool-runs-dry", "off"])
@pytest.mark.parametrize("dim,radius", [(5, 0.06), (16, 1.1)])
def test_radius_capture_any_dimension(gpu, monkeypatch, env, dim, radius):
    for name, value in env.items():  # (an environment switch of the library, or one of its test hooks)
        if name.startswith("PTK_"):
            monkeypatch.setenv(name, value)
        else:
            pt.set_test_knobs(**{name: value})
    pts, q = ds.uniform_cloud(30_000, dim, 91), ds.uniform_cloud(6_000, dim, 92)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 9, device=gpu)
    ref = oracle.Oracle(pts, 9, "port")
    for e in (1.0, 1.4):
        want_off, want = ref.search_radius(q, radius, e=None if e == 1.0 else e)
        got = tree.search_radius(q, radius, e)
        assert np.array_equal(got.offsets, want_off) and got.flat.tobytes() == want.tobytes()
    assert int((np.diff(want_off.astype(np.int64)) > 31).sum()) > 100  # long rows exist: chains grow / break


def test_concurrent_host_threads_on_one_handle(trees):
    """searches may be issued from several host threads on one handle
    (ctypes drops the GIL).  Mixed knn / radius / box calls, every result checked."""
    import threading
    tree, ref, pts, q = trees("uniform")
    parts = [q[i * 4000:(i + 1) * 4000].copy() for i in range(4)]
    want = []
    for p in parts:
        off, flat = ref.search_radius(p, 0.0015)
        want.append((ref.search_knn(p, 1)[:, 0], ref.search_knn(p, 12), off, flat))
    errors = []

    def worker(i):
        try:
            for _ in range(6):
                p, (w1, w12, woff, wflat) = parts[i], want[i]
                if tree.search_knn(p, 1).tobytes() != w1.tobytes():
                    errors.append((i, "knn1"))
                if tree.search_knn(p, 12).tobytes() != w12.tobytes():
                    errors.append((i, "knn12"))
                got = tree.search_radius(p, 0.0015)
                if not np.array_equal(got.offsets, woff) or got.flat.tobytes() != wflat.tobytes():
                    errors.append((i, "radius"))
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_batches_on_several_streams_of_one_handle(trees):
    """k-NN calls arriving on different HIP streams get a scratch block each (up to 4 per handle; a fifth
    stream shares the first block in stream order) and may overlap on the device; every result is checked."""
    import torch
    tree, ref, pts, q = trees("lidar")
    dq = torch.from_numpy(q).cuda()
    want1 = ref.search_knn(q, 1)
    want9 = ref.search_knn(q, 9)
    streams = [torch.cuda.Stream() for _ in range(5)]
    outs = []
    for rep in range(3):
        for i, st in enumerate(streams):
            k = 1 if (i + rep) % 2 == 0 else 9
            with torch.cuda.stream(st):
                outs.append((k, tree.search_knn(dq, k)))
    torch.cuda.synchronize()
    for k, got in outs:
        want = want1 if k == 1 else want9
        assert got.numpy().reshape(want.shape).tobytes() == want.tobytes()


@pytest.mark.parametrize("dim", [2, 3, 6])
def test_box_search_on_device_buffers(gpu, dim):
    """ptk_search_box_count_device / _fill_device: boxes and rows stay on the device."""
    import torch
    pts, q = ds.uniform_cloud(40_000, dim, 71), ds.uniform_cloud(12_000, dim, 72)
    half = np.float32(0.5 * (300.0 / len(pts)) ** (1.0 / dim))
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    ref = oracle.Oracle(pts, 10, "port")
    want_off, want = ref.search_box(q - half, q + half)
    for mode in (pt.REORDER_AUTO, pt.REORDER_OFF):
        tree.set_reorder(mode)
        off, rows = tree.search_box_device(torch.from_numpy(q - half).cuda(), torch.from_numpy(q + half).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off) and want_off[-1] > len(q)
        assert np.array_equal(rows.cpu().numpy(), want)


def test_multi_device_handle_on_the_devices_present(gpu, monkeypatch):
    """ptk_multi_* with every visible device (one on the test box): host form (ranges moved by the
    devices themselves) and device form; with the test hook multi_self_gather=1 devices[0] sends its rows to
    itself, so the grouped ncclSend / ncclRecv path runs on a one-GPU box too."""
    import torch
    pts, q = ds.lidar_cloud(60_000, 1), ds.lidar_cloud(30_001, 2, pose=(3.0, 1.5))
    ref = oracle.Oracle(pts, 10, "port")
    multi = pt.MultiKdTree(pts, 10, devices=list(range(pt.device_count())))
    want1, want8 = ref.search_knn(q, 1)[:, 0], ref.search_knn(q, 8)
    assert multi.search_knn(q, 1).tobytes() == want1.tobytes()
    assert multi.search_knn(q, 8).tobytes() == want8.tobytes()
    got = multi.search_radius(q, 1.0)
    off, flat = ref.search_radius(q, 1.0)
    assert np.array_equal(got.offsets, off) and got.flat.tobytes() == flat.tobytes()
    dq = torch.from_numpy(q).to(f"cuda:{multi.devices[0]}")
    for self_gather in ("0", "1"):
        pt.set_test_knobs(multi_self_gather=int(self_gather))
        for k, want in ((1, want1), (8, want8)):
            rows = multi.search_knn(dq, k).numpy()
            torch.cuda.synchronize()
            assert rows.tobytes() == want.tobytes(), (self_gather, k)


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference not present")
@pytest.mark.parametrize("metric", ["SO2", "SE2Squared"])
def test_topological_metrics_on_the_device(gpu, metric, tmp_path):
    """metric_so2 / metric_se2_squared (metric.hpp:186-257) searched on the device as
    search_nearest_topological does (kd_tree_search.hpp:115-229; four bounds per branch), against
    kd_tree<space, metric_so2 | metric_se2_squared> of the reference's own headers, wrap-around
    cases included (queries next to the seam 0 ~ 1)."""
    if metric == "SO2":
        pts, q, leaf = ds.uniform_cloud(200_000, 1, 51), ds.uniform_cloud(30_000, 1, 52), 10
        q[:200] = np.float32(0.00001) * np.arange(200, dtype=np.float32)[:, None]
        q[200:400] = np.float32(1.0) - np.float32(0.00001) * np.arange(200, dtype=np.float32)[:, None]
        radius = 0.0001
    else:
        pts, q, leaf = ds.uniform_cloud(200_000, 3, 53), ds.uniform_cloud(30_000, 3, 54), 10
        q[:500, 2] = np.float32(0.0005)
        q[500:1000, 2] = np.float32(0.9995)
        radius = 0.0004
    tree = pt.KdTree(pts, pt.Metric[metric], leaf, device=gpu)
    ref = oracle.Oracle(pts, leaf, "reference", metric)
    assert tree.search_knn(q, 1).tobytes() == ref.search_knn(q, 1)[:, 0].tobytes()
    for k in (7, 16, 40):
        assert tree.search_knn(q, k).tobytes() == ref.search_knn(q, k).tobytes()
    assert tree.search_knn(q, 6, 1.3).tobytes() == ref.search_knn(q, 6, e=1.3).tobytes()
    got = tree.search_radius(q, radius)
    off, flat = ref.search_radius(q, radius)
    assert off[-1] > 0 and np.array_equal(got.offsets, off) and got.flat.tobytes() == flat.tobytes()
    got = tree.search_radius(q, radius, 1.5, sort=True)
    off, flat = ref.search_radius(q, radius, e=1.5, sort=True)
    assert np.array_equal(got.offsets, off) and np.array_equal(got.flat["distance"], flat["distance"])
    knn = ref.search_knn(q, 4)
    assert (np.abs(pts[knn["index"], -1] - q[:, -1:]) > 0.5).any()  # neighbours through the seam
    # saved and loaded straight into a device handle (four bounds per branch in the stream)
    path = str(tmp_path / "topological.pkd")
    pt.save_kd_tree(tree, path)
    again = pt.load_kd_tree(pts, path, device=gpu)
    assert again.search_knn(q, 7).tobytes() == ref.search_knn(q, 7).tobytes()
    with pytest.raises(pt.PtkError):  # dimension check
        pt.KdTree(ds.uniform_cloud(100, 2, 1), pt.Metric[metric], 10, device=gpu)


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference not present")
@pytest.mark.parametrize("metric", ["SO2", "SE2Squared"])
def test_topological_box_search_on_the_device(gpu, metric):
    """search_box of a kd_tree<space, metric_so2 | metric_se2_squared> (kd_tree_search.hpp:238-381 with the
    metric_box_map query of box.hpp:300-376): boxes whose circle interval wraps around the seam (min > max) included,
    against the reference's own headers; rows in the reference's report order."""
    rng = np.random.default_rng(77)
    if metric == "SO2":
        pts, leaf, nb = ds.uniform_cloud(100_000, 1, 61), 8, 20_000
        mins = rng.random((nb, 1), dtype=np.float32)
        maxs = (mins + rng.random((nb, 1), dtype=np.float32) * np.float32(0.02)).astype(np.float32)
        wrap = maxs[:, 0] > 1.0                     # intervals through the seam: max comes back below min
        maxs[wrap, 0] -= np.float32(1.0)
        assert wrap.sum() > 50
    else:
        pts, leaf, nb = ds.uniform_cloud(150_000, 3, 63), 10, 20_000
        mins = rng.random((nb, 3), dtype=np.float32)
        maxs = (mins + np.float32(0.03) * (1 + rng.random((nb, 3), dtype=np.float32))).astype(np.float32)
        wrap = maxs[:, 2] > 1.0
        maxs[wrap, 2] -= np.float32(1.0)
        assert wrap.sum() > 50
    tree = pt.KdTree(pts, pt.Metric[metric], leaf, device=gpu)
    ref = oracle.Oracle(pts, leaf, "reference", metric)
    boxes = np.empty((2 * nb, pts.shape[1]), dtype=np.float32)
    boxes[0::2], boxes[1::2] = mins, maxs
    got = tree.search_box(boxes)
    off, flat = ref.search_box(mins, maxs)
    assert off[-1] > 0 and np.array_equal(got.offsets, off) and np.array_equal(got.flat, flat)
    counts = np.diff(off)
    assert counts[wrap].sum() > 0  # the boxes through the seam do find points


def _blind_disc(n, scale=1.0):
    """Queries inside the empty disc under the scanner of the LiDAR-like cloud: the reference's depth-first search of
    such a query visits a long chain of leaves (the expensive queries of BASELINE config 2)."""
    u = ds.raw_uniform24(11, 2 * n).reshape(n, 2)
    r = 12.0 * np.sqrt(u[:, 0])
    a = 2.0 * np.pi * u[:, 1]
    return np.ascontiguousarray(np.stack([r * np.cos(a), r * np.sin(a), np.zeros(n)], axis=1) * scale, dtype=np.float32)


@pytest.mark.parametrize("cloud", ["ties", "self", "blind-disc"])
@pytest.mark.parametrize("direct", ["0", "2"])
def test_capped_phase2_cooperative_search_and_replay_really_run(gpu, monkeypatch, cloud, direct):
    """The only place where the k = 1 search does NOT replay the reference's visit order: with a cap of 1 or 2 far
    children nearly every continuation is handed to the cooperative search; on a lattice cloud (exact ties beyond
    the tie budget) the certificate FAILS for some queries and they come back through the replay; the queries of the
    scanner's blind disc (on a grid of 0.5) are the long chains the cooperative search exists for.  Both forms: the
    ranked classes through phase 2 first (test hook coop_direct=0) and straight from phase 1 on a second stream (2, with
    the HBM spill of the subtree pool).  Counters say the paths ran; rows equal the oracle."""
    import torch

    if cloud == "blind-disc":
        pts = ds.lidar_cloud(400_000, seed=1, unit_scale=20.0)
        q = np.concatenate([_blind_disc(20_000, 20.0), ds.lidar_cloud(20_000, seed=2, pose=(3.0, 1.5), unit_scale=20.0)])
        grid = 0.5
        pts = np.ascontiguousarray(np.round(pts / grid) * grid, dtype=np.float32)
        q = np.ascontiguousarray(np.round(q / grid) * grid, dtype=np.float32)
    else:
        pts, q = _clouds(cloud, 120_000, 40_000)
    pt.set_test_knobs(coop_direct=int(direct))
    if cloud == "ties":  # (the lattice cloud is all piles: on the view without them -- ptk_piles.hpp -- nothing is left to replay)
        pt.set_test_knobs(pile_view=0)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    ref = oracle.Oracle(pts, 10, "port")
    ref.set_threads(ref.max_threads())
    want = ref.search_knn(q, 1)[:, 0]
    dq = torch.from_numpy(q).to(f"cuda:{gpu}")
    redone = 0
    for cap in ("1", "2"):
        pt.set_test_knobs(p2_cap=int(cap))
        got = tree.search_knn(dq, 1).numpy()
        torch.cuda.synchronize()
        counts = tree.knn1_counts()
        assert got.tobytes() == want.tobytes(), cap
        if cloud != "self":  # (a query that IS a tree point is final after phase 1: distance 0)
            assert counts["cooperative"] > 0, counts
        redone += counts["redone"]
    if cloud == "ties":  # more exact ties per query than a lane resolves: the certificate fails, the replay runs
        assert redone > 0
        pt.set_test_knobs(pile_view=None)
        view = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
        assert view.piles()["piles"] > 0 and view.search_knn(dq, 1).numpy().tobytes() == want.tobytes()
        assert view.knn1_counts()["redone"] == 0


def test_k_nearest_cap_follows_the_batch_and_large_batches_run_as_two_launches(gpu, monkeypatch):
    """1 < k <= 32, exact, metric_l2_squared: the far children a query may enter before a wavefront takes it over follow
    the size of the batch (knn_cap of ptk_backend.hip), a full hand-over list leaves a query in its lane, and a batch of
    4 M queries or more goes through as two capped launches side by side.  Rows never depend on any of that: the
    default form, one launch, a fixed cap, no cap at all (the reference traversal in every lane) and a hand-over list of
    almost no entries agree byte for byte on a 4.3 M-query batch, and the small pieces equal the oracle."""
    import torch

    pts, q = ds.config2_clouds("L", 2_000_000, 4_300_000)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    dq = torch.from_numpy(q).to(f"cuda:{gpu}")
    k = 5
    out = torch.empty((len(q), k, 2), dtype=torch.int32, device=dq.device)

    def rows(**env):
        if env:
            pt.set_test_knobs(**env)
        tree.search_knn(dq, k, out)
        torch.cuda.synchronize()
        counts = tree.knn_coop_counts() if env.get("knn_cap") != 0 else None
        pt.set_test_knobs()
        return out.cpu().numpy().tobytes(), counts

    base, counts = rows()
    assert counts["cooperative"] > 0 and counts["redone"] <= counts["cooperative"] // 50, counts
    assert rows(knn_overlap_pct=0)[0] == base        # one capped launch
    assert rows(knn_overlap_pct=50, knn_cap=24)[0] == base
    assert rows(knn_cap=0)[0] == base                 # every query to its end in its lane
    # (a cap of 3 hands most queries over; the list is a 64th of the batch: the rest goes on in its lanes)
    got, counts = rows(knn_cap=3, knn_overlap_pct=0)
    assert got == base and counts["cooperative"] > len(q) // 64, counts
    # pieces of a batch (a shard, a piece of a host-buffer call) get a cap of their own: against the oracle
    ref = oracle.Oracle(pts, 10, "port")
    ref.set_threads(ref.max_threads())
    for n in (5_000, 70_000):
        got = tree.search_knn(dq[:n], 16).numpy()
        torch.cuda.synchronize()
        assert got.tobytes() == ref.search_knn(q[:n], 16).tobytes(), n
        assert tree.knn_coop_counts()["cooperative"] > 0


@pytest.mark.parametrize("cloud,radius,leaf", [("scan", 1.0, 10), ("scan", 6.0, 10), ("ties", 0.05, 10), ("uniform", 0.002, 1),
                                               ("uniform", 0.004, 40)])
def test_radius_search_with_long_queries_finished_by_wavefronts(gpu, cloud, radius, leaf):
    """ptk_kernels_coopr.hpp: the list pass of the radius search capped (test hook radius_cap: 2 and 24 far children per
    query, then the rule of the batch), the queries it hands over counted and filled by a wavefront each in the
    reference's row order, rows it cannot finish (more than 512 leaves with hits) searched again by one lane.  Offsets
    and rows byte-equal to the oracle; exact and approximate; device buffers and host buffers."""
    import torch

    if cloud == "scan":
        pts, q = ds.lidar_cloud(400_000, seed=1, unit_scale=4.0), ds.lidar_cloud(30_000, seed=2, pose=(3.0, 1.5), unit_scale=4.0)
    elif cloud == "ties":
        pts, q = _clouds("ties", 60_000, 20_000)
    else:
        pts, q = ds.uniform_cloud(200_000, 3, 11), ds.uniform_cloud(20_000, 3, 12)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, leaf, device=gpu)
    ref = oracle.Oracle(pts, leaf, "port")
    ref.set_threads(ref.max_threads())
    dq = torch.from_numpy(q).to(f"cuda:{gpu}")
    handed = 0
    for e in (None, 1.5):
        want_off, want = ref.search_radius(q, np.float32(radius), e=e)
        for cap in (2, 24, None):
            pt.set_test_knobs(radius_cap=cap)
            off, raw = tree.search_radius_device(dq, np.float32(radius), e=e or 1.0)
            torch.cuda.synchronize()
            assert np.array_equal(off.cpu().numpy().astype(np.uint64), want_off), (cloud, radius, e, cap)
            assert raw.cpu().numpy().tobytes() == want.tobytes(), (cloud, radius, e, cap)
            if cap == 2:
                handed += tree.radius_coop_counts()["cooperative"]
                got = tree.search_radius(q, np.float32(radius), e) if e else tree.search_radius(q, np.float32(radius))
                assert np.array_equal(got.offsets, want_off) and got.flat.tobytes() == want.tobytes(), (cloud, radius, e, "host")
    pt.set_test_knobs()
    # (the lattice cloud has heaps of coincident points: a tree deeper than a key has bits for runs uncapped)
    assert handed > 0 or cloud == "ties"


def _line_family_case(rng, kind, jitter):
    """One cloud of tools/fuzz_lines.py: points on a line (or a coarse lattice) in 2-D / 3-D, tiny leaves, queries off
    the line or next to tree points -- thousands of points nearly equally far, box distances that drift by rounding."""
    dim = int(rng.choice([2, 3]))
    n = int(rng.choice([3000, 20000, 60000]))
    nq = int(rng.choice([64, 300]))
    leaf = int(rng.choice([1, 2, 5]))
    scale = float(rng.choice([1.0, 37.5, 1e3]))
    if kind == "line":
        pts = ((rng.random((n, 1)) * rng.random((1, dim)) + 0.25) * scale).astype(np.float32)
    else:  # lattice: equal distances by the hundred
        pts = ((np.round(rng.random((n, dim)) * 24) / 24 + 0.25) * scale).astype(np.float32)
    if jitter:  # (no two distances equal: the long searches end in the FIRST sweep's certificate)
        pts = (pts + rng.normal(0, jitter, pts.shape) * scale).astype(np.float32)
    if rng.random() < 0.5:
        q = ((rng.random((nq, 1)) * rng.random((1, dim)) + 0.25) * scale).astype(np.float32)
    else:
        q = (pts[rng.integers(0, n, nq)] + rng.normal(0, 1e-3, (nq, dim)) * scale).astype(np.float32)
    ks = sorted({int(rng.choice([2, 5, 16])), int(rng.choice([24, 32, 33, 48, 56]))})
    return pts, q, leaf, ks


@pytest.mark.parametrize("kind,jitter", [("line", 0.0), ("line", 1e-5), ("lattice", 0.0), ("lattice", 1e-5)])
def test_capped_k_nearest_on_lines_and_lattices_equals_the_compiled_reference(gpu, monkeypatch, kind, jitter):
    """The adversarial family of the cooperative k > 1 search (ptk_kernels_coopk.hpp; profiles/r05_notes.txt item 24,
    profiles/r06_notes.txt item 1): lines and lattices, leaves of 1 / 2 / 5 points, k = 2 .. 56, with and without a
    jitter that removes the equal distances, THE CAP ON FOR EVERY BATCH (test hook knn_cap_min_nq=1) and low, so nearly
    every query is handed over, merged, second-swept or redone.  Byte-equal to the compiled reference
    (oracle/_ref; the restatement where that is absent).  The first case of the line family is the cloud the
    fuzz soak of r05 failed on (seed 802, case 760: 60 000 points, knn = 16 / 32 / 33)."""
    pt.set_test_knobs(knn_cap_min_nq=1)
    how = "reference" if oracle.have_reference() else "port"
    handed = swept = redone = 0
    cases = []
    if kind == "line" and jitter == 0.0:
        rng = np.random.default_rng([802, 760])
        for _ in range(8):
            rng.choice([1, 2])  # (the draws of tools/fuzz_parity.py before the cloud: dim .. metric)
        pts = ((rng.random((60000, 1)) * rng.random((1, 2)) + 0.25) * 37.5).astype(np.float32)
        rng.random()
        q = ((rng.random((64, 1)) * rng.random((1, 2)) + 0.25) * 37.5).astype(np.float32)
        cases.append((pts, q, 1, [16, 32, 33]))
    for case in range(10):
        cases.append(_line_family_case(np.random.default_rng([606, case, int(jitter * 1e7), kind == "line"]), kind, jitter))
    for pts, q, leaf, ks in cases:
        tree = pt.KdTree(pts, pt.Metric.L2Squared, leaf, device=gpu)
        ref = oracle.Oracle(pts, leaf, how)
        for k in ks:
            for cap in ("4", "32"):
                pt.set_test_knobs(knn_cap=int(cap))
                got = tree.search_knn(q, k)
                assert got.tobytes() == ref.search_knn(q, k).tobytes(), (kind, jitter, len(pts), leaf, k, cap)
                c = tree.knn_coop_counts()
                handed += c["cooperative"]
                swept += c["tie_sweeps"]
                redone += c["redone"]
        tree.close()
        ref.close()
    assert handed > 500, handed            # the cap did hand queries over
    if jitter == 0.0:
        assert swept > 0                   # equal distances: second sweeps ran



@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_tiny_batches_with_long_queries_take_the_capped_searches(gpu, dtype):
    """A call of a handful of queries with a long one among them used to pay that query's whole chain (r06 notes item
    20): the radius searches are capped at any batch size, the k-NN searches from 32 queries on.  Queries inside the empty
    disc under the scanner (the long searches of BASELINE config 2), batches of 1 / 7 / 33 / 64: the oracle's rows, and
    the counters say a wavefront finished them."""
    pts = ds.lidar_cloud(300_000, 1).astype(dtype)
    u = ds.raw_uniform24(11, 2 * 64).reshape(64, 2)
    r, a = 12.0 * np.sqrt(u[:, 0]), 2.0 * np.pi * u[:, 1]
    q = np.ascontiguousarray(np.stack([r * np.cos(a), r * np.sin(a), np.zeros(64)], axis=1)).astype(dtype)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    ref = oracle.Oracle(pts, 10, "port", dtype=dtype)
    radius = float(np.median(ref.search_knn(q, 40)["distance"][:, -1]))
    handed_r = handed_k = 0
    for nq in (1, 7, 33, 64):
        qq = q[:nq]
        want_off, want = ref.search_radius(qq, radius)
        got = tree.search_radius(qq, dtype(radius))
        assert np.array_equal(got.offsets, want_off) and np.array_equal(got.flat["index"], want["index"]), nq
        assert np.ascontiguousarray(got.flat["distance"]).tobytes() == np.ascontiguousarray(want["distance"]).tobytes(), nq
        if dtype is np.float64:
            handed_r += tree.knn_coop_counts()["cooperative"]
        else:  # (the counters of a float32 count pass are read between the device passes: the host entry has finished both)
            import torch

            tree.search_radius_device(torch.from_numpy(qq).cuda(), np.float32(radius))
            handed_r += tree.radius_coop_counts()["cooperative"]
        for k in (1, 16):
            w = ref.search_knn(qq, k)
            g = tree.search_knn(qq, k).reshape(nq, k)
            assert np.array_equal(g["index"], w["index"]) and np.ascontiguousarray(g["distance"]).tobytes() == \
                np.ascontiguousarray(w["distance"]).tobytes(), (nq, k)
            if nq >= 32 and (k > 1 or dtype is np.float64):
                handed_k += tree.knn_coop_counts()["cooperative"]
    assert handed_r > 0 and handed_k > 0, (handed_r, handed_k)
    tree.close()
    ref.close()

@pytest.mark.parametrize("cloud", ["lidar", "uniform"])
def test_batches_that_arrive_coherent_are_not_sorted_again(gpu, cloud):
    """The reference walks the rows in the caller's order (_pyco_tree/kd_tree.hpp:128-134); the k = 1 search samples
    the batch and skips its own Morton sort when the order it came in is already coherent.  Generated order (random
    rays) must still be sorted; Morton-presorted, reversed and block-shuffled batches must not; rows always the
    oracle's.  PTK_REORDER_ON sorts whatever comes."""
    import torch

    n, nq = 400_000, 300_000
    pts, q = _clouds(cloud, n, nq)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    ref = oracle.Oracle(pts, 10, "port")
    ref.set_threads(ref.max_threads())
    order = ds.morton_order(q)
    blocks = np.arange(nq).reshape(-1, 3000)[np.random.default_rng(3).permutation(nq // 3000)].ravel()
    batches = {"generated": (q, 1), "presorted": (q[order], 2), "reversed": (q[order[::-1]], 2),
               "block-shuffled": (q[order][blocks], 2)}
    for name, (batch, how) in batches.items():
        batch = np.ascontiguousarray(batch)
        want = ref.search_knn(batch, 1)[:, 0]
        got = tree.search_knn(torch.from_numpy(batch).to(f"cuda:{gpu}"), 1).numpy()
        torch.cuda.synchronize()
        assert got.tobytes() == want.tobytes(), name
        assert tree.batch_order() == how, (name, tree.batch_order())
    tree.set_reorder(pt.REORDER_ON)
    batch = np.ascontiguousarray(q[order])
    got = tree.search_knn(torch.from_numpy(batch).to(f"cuda:{gpu}"), 1).numpy()
    torch.cuda.synchronize()
    assert tree.batch_order() == 1 and got.tobytes() == ref.search_knn(batch, 1)[:, 0].tobytes()


@pytest.mark.parametrize("grid,shift", [(0.1, 0.0), (0.25, 0.3), (1.0, 0.0), (1.0, 0.3), (4.0, 0.0), (4.0, 0.3), (4.0, 0.5)])
def test_coincident_points_k1_through_the_view_without_the_piles(gpu, monkeypatch, grid, shift):
    """Coordinates snapped to a grid (tie-prone data; from 1.0 on piles of coincident points, which the builder peels
    apart one level per point -- at 4.0 a 400 k-point tree is a thousand levels deep): the k = 1 search runs on the
    view in which a pile is one point and reports the point of the pile the reference visits first (ptk_piles.hpp).
    400 k points / 300 k queries on the grid, off the grid, and half a cell off it (several piles at exactly the same
    distance); rows equal the oracle's, exact and approximate, device and host buffers, and the rows of the full tree
    (test hook pile_view=0); on the view next to nothing is replayed lane by lane."""
    import torch

    n, nq = 400_000, 300_000
    pts, q = ds.config2_clouds("L", n, nq)
    p2 = np.ascontiguousarray(np.round(pts / grid) * grid, dtype=np.float32)
    q2 = np.ascontiguousarray(np.round(q / grid) * grid + np.float32(shift * grid), dtype=np.float32)
    tree = pt.KdTree(p2, pt.Metric.L2Squared, 10, device=gpu)
    piles = tree.piles()
    ref = oracle.Oracle(p2, 10, "port")
    ref.set_threads(ref.max_threads())
    want = ref.search_knn(q2, 1)[:, 0]
    dq = torch.from_numpy(q2).to(f"cuda:{gpu}")
    got = tree.search_knn(dq, 1).numpy()
    torch.cuda.synchronize()
    assert got.tobytes() == want.tobytes()
    if grid >= 4.0:
        assert piles["piles"] > 1000 and piles["knn1_depth"] < 64 < tree.info()["max_depth"]
        assert tree.knn1_counts()["redone"] <= 8
    elif grid >= 1.0:
        assert piles["piles"] >= 1 and piles["knn1_depth"] <= tree.info()["max_depth"]
    else:
        assert piles["piles"] == 0 and piles["knn1_depth"] == tree.info()["max_depth"]
    assert tree.search_knn(q2[:50_000], 1).tobytes() == want[:50_000].tobytes()  # host buffers, in pieces
    assert tree.search_knn(dq, 1, 1.25).numpy().tobytes() == ref.search_knn(q2, 1, e=1.25)[:, 0].tobytes()
    if grid == 4.0 and shift == 0.3:  # the same rows the long way: every point of every pile visited
        pt.set_test_knobs(pile_view=0)
        full = pt.KdTree(p2, pt.Metric.L2Squared, 10, device=gpu)
        assert full.piles()["piles"] == 0
        assert full.search_knn(dq[:20_000].contiguous(), 1).numpy().tobytes() == want[:20_000].tobytes()


@pytest.mark.parametrize("form", ["0", "1"])
@pytest.mark.parametrize("nq", [4_097, 70_000, 1_100_000, 2_300_000])
def test_batch_order_is_the_stable_sort_of_the_morton_keys(gpu, monkeypatch, form, nq):
    """The order a batch is searched in (ptk_debug_batch_permutation) is the permutation a stable sort of the Morton
    keys gives -- whichever sort makes it: the library's passes with one wavefront per tile or with blocks of four
    (test hook sort_block), rocprim's above 2 M rows -- compared with the keys of the emulated key kernel sorted by numpy.
    Sizes: a tile and one item, a few tiles, 16-bit and 24-bit keys, a last tile that is partly empty."""
    import torch
    from tests.emu import EmulatedTree

    pt.set_test_knobs(sort_block=int(form))
    pts = ds.lidar_cloud(200_000, seed=1)
    q = ds.lidar_cloud(nq, seed=2, pose=(3.0, 1.5))
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    got = tree.batch_permutation(torch.from_numpy(q).to(f"cuda:{gpu}")).cpu().numpy().astype(np.uint32)
    emu = EmulatedTree(pts, 10)
    want, keys = emu.morton_permutation(q, bits=tree.key_bits(nq))
    assert sorted(got[:: max(1, nq // 5000)].tolist()) == sorted(set(got[:: max(1, nq // 5000)].tolist()))
    assert np.array_equal(keys[got], keys[want])  # sorted by key ...
    assert np.array_equal(got, want)              # ... and stable


def test_rows_in_page_locked_blocks_of_the_pool(gpu):
    """search_knn(pts, k) returns a NEW array per call like the reference's module (def_kd_tree.cpp:73-82); here it is
    built on a page-locked block (ptk_host_alloc) the device writes directly, and the block is handed out again once
    the array is gone.  Rows equal the oracle whichever way the arrays are held: pooled rows, a pageable array of
    the caller, queries in page-locked memory too, no pinned memory at all (test hook host_direct=0)."""
    import gc
    import os

    pts, q = _clouds("lidar", 200_000, 300_000)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    ref = oracle.Oracle(pts, 10, "port")
    ref.set_threads(ref.max_threads())
    want1, want4 = ref.search_knn(q, 1)[:, 0], ref.search_knn(q, 4)
    a = tree.search_knn(q, 1)
    addr_a = a.__array_interface__["data"][0]
    b = tree.search_knn(q, 1)
    assert a.tobytes() == want1.tobytes() and b.tobytes() == want1.tobytes()
    assert addr_a != b.__array_interface__["data"][0]           # two arrays alive: two blocks
    view = a[1000:2000]
    del a
    gc.collect()
    c = tree.search_knn(q, 1)                                    # (the view keeps its block: not handed out again)
    assert c.__array_interface__["data"][0] != addr_a and view.tobytes() == want1[1000:2000].tobytes()
    del view, c
    gc.collect()
    d = tree.search_knn(q, 1)                                    # a freed block comes back
    assert d.tobytes() == want1.tobytes()
    own = np.empty((len(q), 4), dtype=pt.NEIGHBOR)
    assert tree.search_knn(q, 4, own) is own and own.tobytes() == want4.tobytes()
    qp = pt.empty_pinned(q.shape, q.dtype)
    qp[...] = q
    assert tree.search_knn(qp, 4).tobytes() == want4.tobytes()
    pt.set_test_knobs(host_direct=0)
    try:
        assert tree.search_knn(qp, 1).tobytes() == want1.tobytes()
    finally:
        pt.set_test_knobs(host_direct=None)


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference not present")
def test_a_call_the_device_refuses_raises_unless_the_host_loop_is_asked_for(gpu):
    """A topological tree some 3000 levels deep (coincident angles, leaf size 1) is beyond the device stack of the
    topological kernels: ptk_search_* answer PTK_ERR_UNSUPPORTED and the wrapper RAISES -- the batched path has no CPU
    fallback.  Only after pico_tree_amd.allow_host_loop(True) is the call served by the library's host loop
    (ptk_host_search_*: the reference's own batch loop) after ONE warning -- rows equal the reference's
    kd_tree<space, metric_so2>."""
    import warnings

    ring = np.empty((4000, 1), dtype=np.float32)
    ring[:3000, 0] = 0.25
    ring[3000:, 0] = (np.arange(1000) % 997) / np.float32(997.0)
    q = (np.arange(300, dtype=np.float32) / np.float32(300.0)).reshape(-1, 1)
    tree = pt.KdTree(ring, pt.Metric.SO2, 1, device=gpu)
    ref = oracle.Oracle(ring, 1, "reference", "SO2")
    out = np.empty((len(q), 3), dtype=pt.NEIGHBOR)
    assert pt._load().ptk_search_knn(tree._h, q.ctypes.data, len(q), 3, np.float32(1.0), out.ctypes.data) == -2
    with pytest.raises(pt.PtkError, match="too deep"):
        tree.search_knn(q, 3)
    with pytest.raises(pt.PtkError, match="too deep"):
        tree.search_radius(q, 0.001)
    pt.allow_host_loop(True)
    try:
        pt._host_loop_warned = False
        with pytest.warns(RuntimeWarning, match="device search refused"):
            got = tree.search_knn(q, 3)
        assert got.tobytes() == ref.search_knn(q, 3).tobytes()
        with warnings.catch_warnings():
            warnings.simplefilter("error")  # (no second warning)
            rows = tree.search_radius(q, 0.001)
        off, flat = ref.search_radius(q, 0.001)
        assert np.array_equal(rows.offsets, off) and rows.flat.tobytes() == flat.tobytes()
    finally:
        pt.allow_host_loop(False)


def test_config1_through_the_device_matches_the_committed_hashes(gpu):
    """BASELINE configs[0] (100 k / 100 k uniform, knn = 1, leaf 10): the reference's own CPU-runnable case, through
    the HIP path, against tests/golden/hashes.json (SHA-256 of the compiled reference's indices and distance bits)."""
    import hashlib
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hashes.json")) as f:
        e = json.load(f)["config1_uniform_100k_knn1"]
    p, q = ds.uniform_cloud(100_000, 3, seed=1), ds.uniform_cloud(100_000, 3, seed=2)
    tree = pt.KdTree(p, pt.Metric.L2Squared, 10, device=gpu)
    for rows in (tree.search_knn(q, 1), None):
        if rows is None:  # device buffers
            import torch
            rows = tree.search_knn(torch.from_numpy(q).to(f"cuda:{gpu}"), 1).numpy()
        assert hashlib.sha256(np.ascontiguousarray(rows["index"]).tobytes()).hexdigest() == e["index_sha256"]
        assert hashlib.sha256(np.ascontiguousarray(rows["distance"]).tobytes()).hexdigest() == e["distance_bits_sha256"]
        assert int(rows["index"].astype(np.int64).sum()) == e["index_sum"]


def test_callers_arrays_page_locked_in_place(gpu):
    """ptk_host_register / ptk_host_unregister: the caller's own query and result arrays page-locked for a block of
    calls -- same rows as with pageable arrays; a pointer that is not registered is refused."""
    pts, q = ds.lidar_cloud(60_000, 1), ds.lidar_cloud(300_000, 2, pose=(3.0, 1.5))
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=0)
    want = tree.search_knn(q, 1)
    nns = np.empty(len(q), dtype=pt.NEIGHBOR)
    with pt.registered(q), pt.registered(nns):
        tree.search_knn(q, 1, nns)
        assert nns.tobytes() == want.tobytes()
    tree.search_knn(q, 1, nns)  # (pageable again)
    assert nns.tobytes() == want.tobytes()
    assert pt._load().ptk_host_unregister(q.ctypes.data) != 0


@pytest.mark.parametrize("n", [3, 8])
def test_row_cutting_of_the_multi_device_library_with_replicas_on_one_gpu(gpu, monkeypatch, n):
    """The C library's n > 1 path on the one-GPU test box (the loop of _pyco_tree/kd_tree.hpp:117-135 cut into n row
    ranges): with PTK_MULTI_ALLOW_REPLICAS=1 the same device may be listed n times -- n replicas, n host threads / n
    streams with their events and staging, ranges of ceil(nq / n) rows with a ragged (n = 8: EMPTY last ranges are
    covered by the second batch) last one -- and only the transport differs from a node (copies instead of RCCL for
    the device form).  k = 1 / 8 and the radius search against the oracle, nq not divisible by n."""
    import torch

    monkeypatch.setenv("PTK_MULTI_ALLOW_REPLICAS", "1")
    pts, q = ds.lidar_cloud(120_000, 1), ds.lidar_cloud(50_021, 2, pose=(3.0, 1.5))
    ref = oracle.Oracle(pts, 10, "port")
    ref.set_threads(ref.max_threads())
    multi = pt.MultiKdTree(pts, 10, devices=[0] * n)
    assert multi.devices == [0] * n
    for batch in (q, q[:5]):  # (5 rows on 8 entries: ranges of one row, the last three empty)
        want1, want8 = ref.search_knn(batch, 1)[:, 0], ref.search_knn(batch, 8)
        assert multi.search_knn(batch, 1).tobytes() == want1.tobytes()        # host buffers: one thread per entry
        assert multi.search_knn(batch, 8).tobytes() == want8.tobytes()
        dq = torch.from_numpy(batch).to("cuda:0")
        for k, want in ((1, want1), (8, want8)):                             # device buffers: streams, events, staging
            rows = multi.search_knn(dq, k).numpy()
            torch.cuda.synchronize()
            assert rows.reshape(want.shape).tobytes() == want.tobytes(), (n, k, len(batch))
        got = multi.search_radius(batch, 0.02)
        off, flat = ref.search_radius(batch, 0.02)
        assert np.array_equal(got.offsets, off) and got.flat.tobytes() == flat.tobytes()
    monkeypatch.delenv("PTK_MULTI_ALLOW_REPLICAS")
    with pytest.raises(Exception):
        pt.MultiKdTree(pts, 10, devices=[0, 0])  # (without the switch a device listed twice is refused)


def test_all_devices_of_the_node(gpu, monkeypatch):
    """With two or more GPUs visible: ptk_multi_* over ALL of them (real peer send / recv through RCCL, no
    self-gather) and one rank per GPU through pico_tree_amd.sharded over RCCL.  Skips on the one-GPU test box; runs
    the day the suite sees a node."""
    import torch

    n_dev = pt.device_count()
    if n_dev < 2:
        pytest.skip("one GPU visible")
    pts, q = ds.lidar_cloud(200_000, 1), ds.lidar_cloud(100_003, 2, pose=(3.0, 1.5))
    ref = oracle.Oracle(pts, 10, "port")
    ref.set_threads(ref.max_threads())
    want1, want8 = ref.search_knn(q, 1)[:, 0], ref.search_knn(q, 8)
    multi = pt.MultiKdTree(pts, 10, devices=list(range(n_dev)))
    assert multi.search_knn(q, 1).tobytes() == want1.tobytes()
    dq = torch.from_numpy(q).to("cuda:0")
    for k, want in ((1, want1), (8, want8)):
        rows = multi.search_knn(dq, k).numpy()
        torch.cuda.synchronize()
        assert rows.tobytes() == want.tobytes(), k
    # one process per GPU (what bench.py --gpus N runs): torch.distributed over RCCL, rows gathered on rank 0
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_dev}",
           "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(root, "bench.py"), "--gpus", str(n_dev),
           "--steps", "3", "--warmup", "1", "--n", "400000", "--nq", "200003", "--no-cpu-baseline"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    import json
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == n_dev and line["parity_sample_ok"] is True
