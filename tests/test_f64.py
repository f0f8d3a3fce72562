"""kd_trees over float64 points (ptk_tree64_* / ptk_search64_*, pico_tree_amd.KdTree on float64 arrays).

The reference's Python module dispatches on the array dtype
(/root/reference/src/pyco_tree/pico_tree/_pyco_tree/kd_tree.hpp:383-445) and its own unit tests
build float64 trees (test/pyco_tree/kd_tree_test.py:13-18, 209-212, 223-237).

CPU tier: the oracle's double build against the committed goldens (outputs of the COMPILED
REFERENCE over double, tests/golden/make_golden_f64.py) and against the compiled reference itself;
the device kernels run lane by lane in the emulator; the host-only handle (build, save stream).
GPU tier (``-m gpu``): the same checks through the C ABI on the device.
"""

from __future__ import annotations

import ctypes
import os

import numpy as np
import pytest

import oracle
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SETS = ["g_f64_3d", "g_f64_6d"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def same(rows, index, distance):
    return np.array_equal(rows["index"], index) and rows["distance"].tobytes() == distance.tobytes()


def check_against_golden(impl, g, metrics=None):
    """impl: search_knn(q, k, e=None) / search_radius(q, r, e=None, sort=False) -> (offsets, flat) /
    search_box(mins, maxs) -> (offsets, flat); distance BITS are compared."""
    q, k, e, radius = g["queries"], int(g["k"]), float(g["e"]), float(g["radius"])
    assert same(impl.search_knn(q, 1).reshape(len(q), 1), g["knn1_index"], g["knn1_distance"])
    assert same(impl.search_knn(q, k), g["knn_index"], g["knn_distance"])
    assert same(impl.search_knn(q, k, e=e), g["aknn_index"], g["aknn_distance"])
    off, flat = impl.search_radius(q, radius)
    assert np.array_equal(off, g["radius_offsets"]) and same(flat, g["radius_index"], g["radius_distance"])
    off, flat = impl.search_radius(q, radius, sort=True)
    assert np.array_equal(off, g["radius_offsets"])
    assert flat["distance"].tobytes() == g["radius_sorted_distance"].tobytes()
    off, flat = impl.search_radius(q, radius, e=e)
    assert np.array_equal(off, g["aradius_offsets"]) and same(flat, g["aradius_index"], g["aradius_distance"])
    off, flat = impl.search_box(g["box_mins"], g["box_maxs"])
    assert np.array_equal(off, g["box_offsets"]) and np.array_equal(flat, g["box_flat"])
    for metric in ("L1", "LPInf"):  # (the committed goldens hold these two)
        if metrics is not None:
            assert same(metrics(metric).search_knn(q, k), g[f"knn_{metric}_index"], g[f"knn_{metric}_distance"])


# ---- CPU tier ------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", SETS)
def test_oracle_double_reproduces_the_goldens(name):
    g = load(name)
    leaf = int(g["max_leaf_size"])
    port = oracle.Oracle(g["points"], leaf, "port", dtype=np.float64)
    assert port.save_bytes() == g["save_stream"].tobytes()
    check_against_golden(port, g, lambda m: oracle.Oracle(g["points"], leaf, "port", m, dtype=np.float64))


@pytest.mark.skipif(not oracle.have_reference64(), reason="compiled reference (double) not present")
@pytest.mark.parametrize("dim", [1, 2, 3, 4, 8])
def test_oracle_double_equals_compiled_reference(dim):
    rng = np.random.default_rng(100 + dim)
    pts, q = rng.random((3000, dim)), rng.random((500, dim))
    pts[7:19] = pts[7]
    for metric in ("L2Squared", "L1", "LPInf", "LNInf"):
        a = oracle.Oracle(pts, 6, "port", metric, dtype=np.float64)
        b = oracle.Oracle(pts, 6, "reference", metric, dtype=np.float64)
        assert a.save_bytes() == oracle.canonical_stream64(b.save_bytes())
        for k, e in ((1, None), (11, None), (4, 1.7)):
            x, y = a.search_knn(q, k, e=e), b.search_knn(q, k, e=e)
            assert same(x, y["index"], y["distance"])
        r = 0.2 * dim ** 0.5 if metric != "L2Squared" else 0.02 * dim
        (o1, f1), (o2, f2) = a.search_radius(q, r), b.search_radius(q, r)
        assert np.array_equal(o1, o2) and same(f1, f2["index"], f2["distance"]) and o1[-1] > 0
    (o1, f1), (o2, f2) = a.search_box(q - 0.2, q + 0.2), b.search_box(q - 0.2, q + 0.2)
    assert np.array_equal(o1, o2) and np.array_equal(f1, f2)


@pytest.mark.parametrize("name", SETS)
def test_emulated_double_kernels_reproduce_the_goldens(name):
    """ptk_kernels_f64.hpp compiled for the host, run lane by lane (tests/cpp/emulate_kernels.cpp)."""
    from tests import emu

    g = load(name)
    leaf = int(g["max_leaf_size"])
    t = emu.EmulatedTree64(g["points"], leaf)
    assert t.save_bytes() == g["save_stream"].tobytes()
    check_against_golden(t, g, lambda m: emu.EmulatedTree64(g["points"], leaf, m))
    # the k-list kept in the output row (k > 16 on the device) instead of in registers
    assert same(t.search_knn(g["queries"], int(g["k"]), list_in_registers=False), g["knn_index"], g["knn_distance"])
    assert same(t.search_knn(g["queries"], 3), g["knn_index"][:, :3], g["knn_distance"][:, :3])  # K = 4 registers
    # K = 32 registers (17 <= k <= 32), against the oracle's double build
    ref = oracle.Oracle(g["points"], leaf, "port", dtype=np.float64)
    want = ref.search_knn(g["queries"], 23)
    assert same(t.search_knn(g["queries"], 23), want["index"], want["distance"])


def rows_equal(a, b):
    """Index and distance BITS (the four padding bytes of a neighbor<int, double> are nobody's)."""
    return np.array_equal(a["index"], b["index"]) and a["distance"].tobytes() == b["distance"].tobytes()


def _without_padding(stream: bytes, n: int, dim: int) -> bytes:
    """A four-bound double stream with the 4 padding bytes of every kd_tree_branch_double record (between its int and
    its first double: written as they lie in memory, kd_tree_node.hpp:52-67) zeroed."""
    b = bytearray(stream)
    at = 16 + 4 * n + 16 * dim
    while at < len(b):
        leaf = b[at]
        at += 1
        if leaf:
            at += 8
        else:
            b[at + 4:at + 8] = bytes(4)
            at += 40
    assert at == len(b)
    return bytes(b)


def _topological_case(metric, n, nq, seed):
    """Points, queries (some next to the seam 0 ~ 1 of the circle axis), a radius and boxes (some through the seam)."""
    rng = np.random.default_rng(seed)
    dim = 1 if metric == "SO2" else 3
    pts, q = rng.random((n, dim)), rng.random((nq, dim))
    q[:nq // 20, -1] = 1e-5 * np.arange(nq // 20)
    q[nq // 20:nq // 10, -1] = 1.0 - 1e-5 * np.arange(nq // 10 - nq // 20)
    radius = (30.0 / n) if metric == "SO2" else (0.8 * (30.0 / n) ** (1.0 / 3.0)) ** 2
    nb = max(nq // 2, 50)
    mins = rng.random((nb, dim))
    maxs = mins + (60.0 / n if metric == "SO2" else 0.06) * (1 + rng.random((nb, dim)))
    wrap = maxs[:, -1] > 1.0  # intervals through the seam: max comes back below min
    maxs[wrap, -1] -= 1.0
    assert wrap.sum() > 3
    return pts, q, radius, mins, maxs, wrap


@pytest.mark.skipif(not oracle.have_reference64(), reason="compiled reference (double) not present")
@pytest.mark.parametrize("metric", ["SO2", "SE2Squared"])
def test_emulated_double_topological_kernels_equal_the_compiled_reference(metric):
    """metric_so2 / metric_se2_squared over double points (metric.hpp:186-257) through traverse64_topo and
    box64_kernel<.., TOPO> of ptk_kernels_f64.hpp, against kd_tree<space of double points, metric_so2 |
    metric_se2_squared> of the reference's own headers (oracle/_ref): neighbours and boxes through the seam included."""
    from tests import emu

    pts, q, radius, mins, maxs, wrap = _topological_case(metric, 6_000 if metric == "SO2" else 20_000, 1_200, 71)
    leaf = 6 if metric == "SO2" else 10
    t = emu.EmulatedTree64(pts, leaf, metric)
    ref = oracle.Oracle(pts, leaf, "reference", metric, dtype=np.float64)
    # the four-bound stream (kd_tree_branch_double records)
    assert _without_padding(t.save_bytes(), len(pts), pts.shape[1]) == _without_padding(ref.save_bytes(), len(pts), pts.shape[1])
    for k, reg in ((1, True), (3, True), (7, True), (7, False), (40, True)):
        assert rows_equal(t.search_knn(q, k, list_in_registers=reg), ref.search_knn(q, k)), (k, reg)
    assert rows_equal(t.search_knn(q, 5, e=1.4), ref.search_knn(q, 5, e=1.4))
    for kw in ({}, {"e": 1.5}):
        a, b = t.search_radius(q, radius, **kw), ref.search_radius(q, radius, **kw)
        assert b[0][-1] > 0 and np.array_equal(a[0], b[0]) and rows_equal(a[1], b[1])
    (o1, f1), (o2, f2) = t.search_box(mins, maxs), ref.search_box(mins, maxs)
    assert o2[-1] > 0 and np.array_equal(o1, o2) and np.array_equal(f1, f2)
    assert np.diff(o2)[wrap].sum() > 0  # the boxes through the seam do find points
    knn = ref.search_knn(q, 4)
    assert (np.abs(pts[knn["index"], -1] - q[:, -1:]) > 0.5).any()  # some neighbours are nearer through 0 ~ 1



def _blind_disc_queries64(n):
    """Queries inside the empty disc under the scanner of cloud L (the long searches of BASELINE config 2)."""
    u = ds.raw_uniform24(11, 2 * n).reshape(n, 2)
    r = 12.0 * np.sqrt(u[:, 0])
    a = 2.0 * np.pi * u[:, 1]
    return np.ascontiguousarray(np.stack([r * np.cos(a), r * np.sin(a), np.zeros(n)], axis=1))


def _capped_cases64():
    yield "lidar", ds.lidar_cloud(20_000, 1).astype(np.float64), np.concatenate(
        [ds.lidar_cloud(100, 2, pose=(3.0, 1.5)).astype(np.float64), _blind_disc_queries64(60)]), 10
    yield "ties", (np.round(ds.uniform_cloud(3_000, 3, 6) * 16) / 16).astype(np.float64), \
        (np.round(ds.uniform_cloud(100, 3, 7) * 16) / 16).astype(np.float64), 10
    rng = np.random.default_rng(5)
    line = ((rng.random((6_000, 1)) * rng.random((1, 2)) + 0.25) * 37.5).astype(np.float32).astype(np.float64)
    yield "line", line, ((rng.random((80, 1)) * rng.random((1, 2)) + 0.25) * 37.5).astype(np.float32).astype(np.float64), 1
    yield "dim1", ds.uniform_cloud(3_000, 1, 8).astype(np.float64), ds.uniform_cloud(100, 1, 9).astype(np.float64), 4


@pytest.mark.parametrize("case", list(_capped_cases64()), ids=lambda c: c[0])
def test_emulated_double_capped_knn_and_its_cooperative_search_equal_oracle(case):
    """ptk_kernels_coop64.hpp in the emulator: the capped double kernel (lanes one after the other), the cooperative
    search of what it handed over (64 fibers per wavefront), the reference search of what that could not certify --
    k = 1 included (no two-phase search in double), both metrics it is shipped for, a cap of 0 / 2 far children and a
    small spill (the redo path); equal distances (a lattice, points on a line) must come out in the reference's order."""
    name, pts, q, leaf = case
    for metric in (("L2Squared", "L1") if name in ("lidar", "ties") else ("L2Squared",)):
        from tests import emu

        t = emu.EmulatedTree64(pts, leaf, metric)
        ref = oracle.Oracle(pts, leaf, "port", metric, dtype=np.float64)
        handed = sweeps = 0
        for k in ((1, 3, 16, 32) if metric == "L2Squared" else (1, 16)):
            want = ref.search_knn(q, k)
            for cap, small in ((0, False), (2, True)):
                got, nh, nr = t.search_knn_capped(q, k, cap, pool_small=small)
                assert same(got, want["index"], want["distance"]), (name, metric, k, cap)
                handed += nh
                sweeps += t.last_tie_sweeps
        assert handed > 0, (name, metric)
        if name == "ties":  # (in double the distances along a line of float32 coordinates rarely coincide)
            assert sweeps > 0, (name, metric)



def _radius_for(ref, q, hits):
    """A radius (in the metric's own unit) under which a query has about `hits` neighbours: the median distance of the
    hits-th nearest."""
    return float(np.median(ref.search_knn(q[:50], hits)["distance"][:, -1]))


@pytest.mark.parametrize("case", list(_capped_cases64()), ids=lambda c: c[0])
def test_emulated_double_capped_radius_and_its_cooperative_count_equal_oracle(case):
    """The radius half of ptk_kernels_coop64.hpp in the emulator, both passes: the capped count launch, the cooperative
    count (keyed leaf entries, sorted), the recount of what that lost; the capped fill launch, the cooperative replay,
    the refill.  Every metric of the double kernels (no certificate is involved), exact and approximate, radii of ~60
    and ~600 neighbours (the second overflows the 512 entries of many queries: the redo path), a small spill."""
    from tests import emu

    name, pts, q, leaf = case
    q = q[:120]
    for metric in (("L2Squared", "L1", "LPInf", "LNInf") if name == "lidar" else ("L2Squared", "LNInf")):
        t = emu.EmulatedTree64(pts, leaf, metric)
        ref = oracle.Oracle(pts, leaf, "port", metric, dtype=np.float64)
        handed = redone = 0
        for hits, e in ((60, None), (60, 1.3), (600, None)):
            if hits >= len(pts):
                continue
            r = _radius_for(ref, q, hits)
            woff, wflat = ref.search_radius(q, r, e=e)
            for cap, small in ((0, False), (3, True)):
                off, flat, nh, nr = t.search_radius_capped(q, r, cap, e=e, small=small)
                assert np.array_equal(off, woff) and same(flat, wflat["index"], wflat["distance"]), (name, metric, hits, e, cap)
                handed += nh
                redone += nr
        assert handed > 0 and handed > redone, (name, metric, handed, redone)


def test_host_only_double_handle_builds_the_reference_tree(tmp_path):
    g = load("g_f64_3d")
    t = pt.KdTree(g["points"], pt.Metric.L2Squared, int(g["max_leaf_size"]), device=pt.PTK_DEVICE_NONE)
    assert t.dtype_scalar == np.float64 and t.dtype_neighbor == pt.NEIGHBOR64 and "dtype=float64" in repr(t)
    assert t.dtype_neighbor.itemsize == 16 and t.dtype_neighbor.fields["distance"][1] == 8
    assert t._serialize() == g["save_stream"].tobytes()
    assert t.info()["n_points"] == len(g["points"]) and t.metric(-2.0) == 4.0
    with pytest.raises(pt.PtkError) as err:  # no device replica: loud, no CPU search
        t.search_knn(g["queries"], 1)
    assert err.value.status == -3
    with pytest.raises(ValueError):  # float32 queries on a float64 tree
        t.search_knn(g["queries"].astype(np.float32), 1)
    # file round trip (PKD container around the double stream)
    fn = str(tmp_path / "t64.bin")
    pt.save_kd_tree(t, fn)
    t2 = pt.load_kd_tree(g["points"], fn, device=pt.PTK_DEVICE_NONE)
    assert t2.dtype_scalar == np.float64 and t2._serialize() == t._serialize()
    with pytest.raises(pt.PtkError):  # a float32 stream is not a float64 one
        f32 = pt.KdTree(g["points"].astype(np.float32), pt.Metric.L2Squared, 10, device=pt.PTK_DEVICE_NONE)
        pt.save_kd_tree(f32, fn)
        pt.load_kd_tree(g["points"], fn, device=pt.PTK_DEVICE_NONE)


def test_double_entry_points_validate_arguments():
    lib = pt._load()
    h = ctypes.c_void_p()
    pts = np.random.default_rng(1).random((100, 3))
    none = pt.PTK_DEVICE_NONE
    assert lib.ptk_tree64_create_from_points(None, 100, 3, 10, none, ctypes.byref(h)) == -1
    assert lib.ptk_tree64_create_from_points(pts.ctypes.data, 0, 3, 10, none, ctypes.byref(h)) == -1
    assert lib.ptk_tree64_create_from_points(pts.ctypes.data, 100, 3, 0, none, ctypes.byref(h)) == -1
    assert lib.ptk_tree64_create_from_points(pts.ctypes.data, 100, 3, 10, none, ctypes.byref(h)) == 0
    assert lib.ptk_tree64_set_metric(h, 4) == -1 and lib.ptk_tree64_set_metric(h, 1) == 0
    out = np.zeros(100, dtype=pt.NEIGHBOR64)
    assert lib.ptk_search64_knn(h, pts.ctypes.data, 100, 1, 1.0, out.ctypes.data) == -3  # no device
    assert lib.ptk_search64_knn(None, pts.ctypes.data, 100, 1, 1.0, out.ctypes.data) == -1
    lib.ptk_tree64_destroy(h)
    garbage = ctypes.create_string_buffer(b"\x03" + b"\0" * 40, 41)
    assert lib.ptk_tree64_create_from_stream(pts.ctypes.data, 100, 3, garbage, 41, none, ctypes.byref(h)) == -1


@pytest.mark.skipif(not oracle.have_reference64(), reason="compiled reference (double) not present")
def test_host_only_double_topological_handle(tmp_path):
    """The four-bound tree of a topological space over double points: built by the product's builder, written as
    kd_tree<space, metric_se2_squared>::save writes it, loaded back; the dimension and outer-bound checks."""
    pts = np.random.default_rng(5).random((3_000, 3))
    none = pt.PTK_DEVICE_NONE
    t = pt.KdTree(pts, pt.Metric.SE2Squared, 10, device=none)
    ref = oracle.Oracle(pts, 10, "reference", "SE2Squared", dtype=np.float64)
    assert _without_padding(t._serialize(), 3_000, 3) == _without_padding(ref.save_bytes(), 3_000, 3)
    fn = str(tmp_path / "se2_64.bin")
    pt.save_kd_tree(t, fn)
    t2 = pt.load_kd_tree(pts, fn, device=none)
    assert t2.metric_string == "SE2Squared" and t2._serialize() == t._serialize() and t2.metric(-2.0) == 4.0
    for metric, bad in (("SO2", pts), ("SE2Squared", pts[:, :2].copy())):  # metric_so2: dim 1, metric_se2_squared: dim 3
        with pytest.raises(pt.PtkError) as err:
            pt.KdTree(bad, pt.Metric[metric], 10, device=none)
        assert err.value.status == -1
    # a tree read from a euclidean stream has two bounds per branch: not a topological tree
    lib, h = pt._load(), ctypes.c_void_p()
    plain = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=none)._serialize()
    buf = ctypes.create_string_buffer(plain, len(plain))
    assert lib.ptk_tree64_create_from_stream(pts.ctypes.data, 3_000, 3, buf, len(plain), none, ctypes.byref(h)) == 0
    size = ctypes.c_uint64()
    assert lib.ptk_tree64_set_metric(h, 5) == -1 and lib.ptk_tree64_serialize_topological(h, None, 0, ctypes.byref(size)) == -1
    assert lib.ptk_tree64_serialize(h, None, 0, ctypes.byref(size)) == 0 and size.value == len(plain)
    lib.ptk_tree64_destroy(h)
    assert lib.ptk_tree64_create_from_topological_stream(pts.ctypes.data, 3_000, 3, buf, len(plain), none, ctypes.byref(h)) == -1


# ---- GPU tier ------------------------------------------------------------------------------------

class _GpuTree:
    """Adapts pico_tree_amd.KdTree to the (offsets, flat) interface of check_against_golden."""

    def __init__(self, pts, leaf, metric="L2Squared", device=0):
        self.t = pt.KdTree(pts, pt.Metric[metric], leaf, device=device)

    def search_knn(self, q, k, e=None):
        r = self.t.search_knn(q, k) if e is None else self.t.search_knn(q, k, e)
        return r.reshape(len(q), k)

    def search_radius(self, q, r, e=None, sort=False):
        d = self.t.search_radius(q, r, sort=sort) if e is None else self.t.search_radius(q, r, e, sort=sort)
        return d.offsets, d.flat

    def search_box(self, mins, maxs):
        boxes = np.empty((2 * len(mins), mins.shape[1]), dtype=np.float64)
        boxes[0::2], boxes[1::2] = mins, maxs
        d = self.t.search_box(boxes)
        return d.offsets, d.flat


@pytest.mark.gpu
@pytest.mark.parametrize("name", SETS)
def test_gpu_double_reproduces_the_goldens(gpu, name):
    g = load(name)
    leaf = int(g["max_leaf_size"])
    t = _GpuTree(g["points"], leaf, device=gpu)
    assert t.t._serialize() == g["save_stream"].tobytes()
    check_against_golden(t, g, lambda m: _GpuTree(g["points"], leaf, m, device=gpu))


@pytest.mark.gpu
@pytest.mark.parametrize("dim,n,nq,leaf", [(3, 200_000, 100_000, 10), (2, 50_000, 30_000, 1), (5, 60_000, 20_000, 12),
                                           (16, 20_000, 4_000, 8), (1, 5_000, 3_000, 3)])
def test_gpu_double_equals_oracle(gpu, dim, n, nq, leaf):
    """Seeded clouds at sizes the oracle finishes in seconds; every metric; indices and distance bits."""
    rng = np.random.default_rng(dim * 7 + 1)
    pts, q = rng.random((n, dim)) * 50.0, rng.random((nq, dim)) * 50.0
    pts[100:130] = pts[100]
    for metric in ("L2Squared", "L1", "LPInf", "LNInf"):
        ref = oracle.Oracle(pts, leaf, "port", metric, dtype=np.float64)
        t = _GpuTree(pts, leaf, metric, device=gpu)
        for k, e in ((1, None), (16, None), (40, None), (5, 1.5)):
            want = ref.search_knn(q, k, e=e)
            assert same(t.search_knn(q, k, e=e), want["index"], want["distance"]), (metric, k, e)
        if metric == "L2Squared":
            r = {1: 1e-5, 2: 0.05, 3: 1.0, 5: 60.0, 16: 1500.0}[dim]
        else:
            r = {1: 3e-3, 2: 0.2, 3: 1.0, 5: 8.0, 16: 30.0}[dim] * (2.0 if metric == "L1" else 1.0)
        (o1, f1), (o2, f2) = t.search_radius(q, r), ref.search_radius(q, r)
        assert np.array_equal(o1, o2) and same(f1, f2["index"], f2["distance"]) and o1[-1] > 0, (metric, "radius")
        (o1, f1), (o2, f2) = t.search_radius(q[:2000], r, e=1.2, sort=True), ref.search_radius(q[:2000], r, e=1.2, sort=True)
        assert np.array_equal(o1, o2) and f1["distance"].tobytes() == f2["distance"].tobytes()
    h = 50.0 * 0.5 * (200.0 / n) ** (1.0 / dim)
    (o1, f1), (o2, f2) = t.search_box(q[:5000] - h, q[:5000] + h), ref.search_box(q[:5000] - h, q[:5000] + h)
    assert np.array_equal(o1, o2) and np.array_equal(f1, f2) and o1[-1] > 0


@pytest.mark.gpu
@pytest.mark.skipif(not oracle.have_reference64(), reason="compiled reference (double) not present")
@pytest.mark.parametrize("metric", ["SO2", "SE2Squared"])
def test_gpu_double_topological_metrics(gpu, metric, tmp_path):
    """kd_tree<space of double points, metric_so2 | metric_se2_squared> (metric.hpp:186-257) on the device
    (traverse64_topo, box64_kernel<.., TOPO>) against the reference's own headers compiled over double: knn, the
    approximate search, radius, box -- neighbours and boxes through the seam 0 ~ 1 included -- and the file round trip."""
    import torch

    n, nq = (200_000, 30_000) if metric == "SO2" else (300_000, 30_000)
    pts, q, radius, mins, maxs, wrap = _topological_case(metric, n, nq, 97)
    leaf = 10
    t = _GpuTree(pts, leaf, metric, device=gpu)
    ref = oracle.Oracle(pts, leaf, "reference", metric, dtype=np.float64)
    assert _without_padding(t.t._serialize(), n, pts.shape[1]) == _without_padding(ref.save_bytes(), n, pts.shape[1])
    for k, e in ((1, None), (4, None), (7, None), (16, None), (40, None), (6, 1.3)):
        assert rows_equal(t.search_knn(q, k, e=e), ref.search_knn(q, k, e=e)), (k, e)
    (o1, f1), (o2, f2) = t.search_radius(q, radius), ref.search_radius(q, radius)
    assert o2[-1] > nq and np.array_equal(o1, o2) and rows_equal(f1, f2)
    (o1, f1), (o2, f2) = t.search_radius(q[:3000], radius, e=1.5, sort=True), ref.search_radius(q[:3000], radius, e=1.5, sort=True)
    assert np.array_equal(o1, o2) and f1["distance"].tobytes() == f2["distance"].tobytes()
    (o1, f1), (o2, f2) = t.search_box(mins, maxs), ref.search_box(mins, maxs)
    assert o2[-1] > 0 and np.array_equal(o1, o2) and np.array_equal(f1, f2)
    assert np.diff(o2)[wrap].sum() > 0  # the boxes through the seam do find points
    knn = ref.search_knn(q, 4)
    assert (np.abs(pts[knn["index"], -1] - q[:, -1:]) > 0.5).any()  # neighbours through the seam
    # device tensors in, device tensors out
    dq = torch.from_numpy(q).to(f"cuda:{gpu}")
    got = t.t.search_knn(dq, 7)
    torch.cuda.synchronize()
    assert rows_equal(got.numpy(), ref.search_knn(q, 7))
    # saved and loaded straight into a device handle (four bounds per branch in the stream)
    fn = str(tmp_path / "topological64.pkd")
    pt.save_kd_tree(t.t, fn)
    again = pt.load_kd_tree(pts, fn, device=gpu)
    assert rows_equal(again.search_knn(q, 7).reshape(nq, 7), ref.search_knn(q, 7))
    # a euclidean metric on the same handle afterwards, and back
    lib = pt._load()
    assert lib.ptk_tree64_set_metric(t.t._h, 0) == 0 and lib.ptk_tree64_set_metric(t.t._h, 4 if metric == "SO2" else 5) == 0
    assert rows_equal(t.search_knn(q[:1000], 3), ref.search_knn(q[:1000], 3))


@pytest.mark.gpu
def test_gpu_double_device_tensors_and_python_api(gpu, tmp_path):
    import torch

    g = load("g_f64_3d")
    t = pt.KdTree(g["points"], pt.Metric.L2Squared, int(g["max_leaf_size"]), device=gpu)
    k = int(g["k"])
    dq = torch.from_numpy(g["queries"]).to(f"cuda:{gpu}")
    got = t.search_knn(dq, k)
    torch.cuda.synchronize()
    assert got.raw.dtype == torch.int64 and tuple(got.raw.shape) == (len(g["queries"]), k, 2)
    assert same(got.numpy(), g["knn_index"], g["knn_distance"])
    assert np.array_equal(got.index.cpu().numpy(), g["knn_index"])
    assert got.distance.cpu().numpy().tobytes() == g["knn_distance"].tobytes()
    # kd_tree_test.py:203-229 with a float64 tree: DArray of the tree's neighbor dtype, reused
    d = pt.DArray(dtype=t.dtype_neighbor)
    assert d.dtype == t.dtype_neighbor and not d
    t.search_radius(g["queries"], float(g["radius"]), d)
    assert len(d) == len(g["queries"]) and same(d.flat, g["radius_index"], g["radius_distance"])
    with pytest.raises(ValueError):
        t.search_radius(g["queries"], float(g["radius"]), pt.DArray(pt.NEIGHBOR))
    # kd_tree_test.py:231-248 with float64 points
    fn = str(tmp_path / "tree64.bin")
    pt.save_kd_tree(t, fn)
    t2 = pt.load_kd_tree(g["points"], fn, device=gpu)
    assert repr(t) == repr(t2) and t.dtype_scalar == t2.dtype_scalar
    assert same(t2.search_knn(g["queries"], k), g["knn_index"], g["knn_distance"])
    # a stream written by the reference side (oracle, pinned to the compiled reference) loads too
    ref = oracle.Oracle(g["points"], int(g["max_leaf_size"]), "port", dtype=np.float64)
    with open(fn, "wb") as f:
        f.write(b"\x89PKD" + np.uint32(1).tobytes() + np.uint64(9).tobytes() + b"L2Squared" + ref.save_bytes())
    t3 = pt.load_kd_tree(g["points"], fn, device=gpu)
    assert same(t3.search_knn(g["queries"], 1).reshape(-1, 1), g["knn1_index"], g["knn1_distance"])
    # column-major queries: the transposed (k, nq) result of the reference (kd_tree.hpp:362-378)
    qf = np.asfortranarray(g["queries"].T)
    rf = t.search_knn(qf, k)
    assert rf.shape == (k, len(g["queries"])) and np.array_equal(rf.reshape(-1)["index"], g["knn_index"].reshape(-1))



@pytest.mark.gpu
@pytest.mark.parametrize("case", list(_capped_cases64()) + [("big", None, None, 10)], ids=lambda c: c[0])
def test_gpu_double_capped_knn_and_its_cooperative_search(gpu, case):
    """The capped double k-NN launch + knn64_coop_kernel + knn64_redo_kernel on the device (every exact k <= 32 search of
    256 queries or more with dim <= 3 under metric_l2_squared / metric_l1 takes it): a low cap forced through the test
    hooks so that nearly every query is handed over, the rule's own cap, and no cap -- all three the oracle's rows."""
    name, pts, q, leaf = case
    if name == "big":
        pts = ds.lidar_cloud(300_000, 1).astype(np.float64)
        q = np.concatenate([ds.lidar_cloud(40_000, 2, pose=(3.0, 1.5)).astype(np.float64), _blind_disc_queries64(4_000)])
    else:
        q = np.concatenate([q] * 4)
    for metric in ("L2Squared", "L1"):
        ref = oracle.Oracle(pts, leaf, "port", metric, dtype=np.float64)
        ref.set_threads(os.cpu_count() or 1)
        t = _GpuTree(pts, leaf, metric, device=gpu)
        handed = 0
        for k in (1, 4, 16, 32):
            want = ref.search_knn(q, k)
            for knobs in ({"knn64_cap": 1}, {}, {"knn64_cap": 0}):
                pt.set_test_knobs(**knobs)
                try:
                    got = t.search_knn(q, k)
                    c = t.t.knn_coop_counts()
                finally:
                    pt.set_test_knobs(**{n: None for n in knobs})
                assert same(got, want["index"], want["distance"]), (name, metric, k, knobs)
                if knobs.get("knn64_cap") == 1:
                    handed += c.get("cooperative", 0)
                if knobs.get("knn64_cap") == 0:
                    assert c.get("cooperative", 0) == 0
        assert handed > 0, (name, metric)


def _line_family_case64(rng, kind, jitter):
    """_line_family_case of tests/test_gpu_parity.py in double: points on a line (or a coarse lattice) in 2-D / 3-D, tiny
    leaves, queries off the line or next to tree points -- thousands of points nearly (or, from float32 coordinates and
    on a lattice, exactly) equally far, box distances that drift by rounding."""
    dim = int(rng.choice([2, 3]))
    n = int(rng.choice([3000, 20000, 60000]))
    nq = int(rng.choice([64, 300]))
    leaf = int(rng.choice([1, 2, 5]))
    scale = float(rng.choice([1.0, 37.5, 1e3]))
    if kind == "line":
        pts = (rng.random((n, 1)) * rng.random((1, dim)) + 0.25) * scale
    else:  # lattice: equal distances by the hundred
        pts = (np.round(rng.random((n, dim)) * 24) / 24 + 0.25) * scale
    if rng.random() < 0.5:  # (coordinates that are float32 numbers: their differences and squares are exact in double)
        pts = pts.astype(np.float32).astype(np.float64)
    if jitter:
        pts = pts + rng.normal(0, jitter, pts.shape) * scale
    if rng.random() < 0.5:
        q = (rng.random((nq, 1)) * rng.random((1, dim)) + 0.25) * scale
    else:
        q = pts[rng.integers(0, n, nq)] + rng.normal(0, 1e-3, (nq, dim)) * scale
    ks = sorted({1, int(rng.choice([2, 5, 16])), int(rng.choice([24, 32]))})
    return np.ascontiguousarray(pts), np.ascontiguousarray(q), leaf, ks


@pytest.mark.gpu
@pytest.mark.parametrize("kind,jitter", [("line", 0.0), ("line", 1e-9), ("lattice", 0.0), ("lattice", 1e-9)])
def test_gpu_double_capped_knn_on_lines_and_lattices(gpu, kind, jitter):
    """The adversarial family of the cooperative search (test_capped_k_nearest_on_lines_and_lattices_equals_the_compiled_
    reference of tests/test_gpu_parity.py) for ptk_kernels_coop64.hpp: THE CAP ON FOR EVERY BATCH and low, so that nearly
    every query is handed over, merged, second-swept or redone; k = 1 .. 32; equal to the compiled reference over double
    (oracle/_ref; the restatement where that is absent)."""
    how = "reference" if oracle.have_reference64() else "port"
    handed = swept = 0
    pt.set_test_knobs(knn_cap_min_nq=1)
    try:
        for case in range(8):
            pts, q, leaf, ks = _line_family_case64(np.random.default_rng([707, case, int(jitter * 1e10), kind == "line"]), kind, jitter)
            tree = pt.KdTree(pts, pt.Metric.L2Squared, leaf, device=gpu)
            ref = oracle.Oracle(pts, leaf, how, dtype=np.float64)
            for k in ks:
                want = ref.search_knn(q, k)
                for cap in (2, 24):
                    pt.set_test_knobs(knn64_cap=cap)
                    got = tree.search_knn(q, k).reshape(len(q), k)
                    assert same(got, want["index"], want["distance"]), (kind, jitter, case, len(pts), leaf, k, cap)
                    c = tree.knn_coop_counts()
                    handed += c["cooperative"]
                    swept += c["tie_sweeps"]
            tree.close()
            ref.close()
    finally:
        pt.set_test_knobs(knn_cap_min_nq=None, knn64_cap=None)
    assert handed > 500, handed  # the cap did hand queries over
    if jitter == 0.0 and kind == "lattice":
        assert swept > 0         # equal distances: second sweeps ran


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(_capped_cases64()) + [("big", None, None, 10)], ids=lambda c: c[0])
def test_gpu_double_capped_radius_and_its_cooperative_count(gpu, case):
    """The capped double radius search on the device (every call of 256 queries or more with dim <= 3 under a
    non-topological metric takes it): a low cap forced through the test hooks so that nearly every query is handed over,
    the rule's own cap, and no cap -- all three the oracle's offsets and rows; exact, approximate and sorted."""
    name, pts, q, leaf = case
    if name == "big":
        pts = ds.lidar_cloud(300_000, 1).astype(np.float64)
        q = np.concatenate([ds.lidar_cloud(20_000, 2, pose=(3.0, 1.5)).astype(np.float64), _blind_disc_queries64(3_000)])
    else:
        q = np.concatenate([q] * 4)
    for metric in ("L2Squared", "L1", "LPInf", "LNInf"):
        ref = oracle.Oracle(pts, leaf, "port", metric, dtype=np.float64)
        ref.set_threads(os.cpu_count() or 1)
        t = _GpuTree(pts, leaf, metric, device=gpu)
        handed = 0
        for hits, e, sort in ((60, None, False), (40, 1.3, False), (700, None, True)):
            if hits >= len(pts) or (name == "big" and metric != "L2Squared" and hits > 60):
                continue
            r = _radius_for(ref, q, hits)
            woff, wflat = ref.search_radius(q, r, e=e, sort=sort)
            for knobs in ({"radius64_cap": 1}, {}, {"radius64_cap": 0}):
                pt.set_test_knobs(**knobs)
                try:
                    off, flat = t.search_radius(q, r, e=e, sort=sort)
                    c = t.t.knn_coop_counts()
                finally:
                    pt.set_test_knobs(**{n: None for n in knobs})
                assert np.array_equal(off, woff), (name, metric, hits, e, knobs)
                if sort:  # (equal distances may come in either order of their indices: search_visitor.hpp sorts by distance)
                    assert flat["distance"].tobytes() == wflat["distance"].tobytes(), (name, metric, hits, e, knobs)
                else:
                    assert same(flat, wflat["index"], wflat["distance"]), (name, metric, hits, e, knobs)
                if knobs.get("radius64_cap") == 1:
                    handed += c["cooperative"]
                if knobs.get("radius64_cap") == 0:
                    assert c["cooperative"] == 0
        assert handed > 0, (name, metric)

@pytest.mark.gpu
def test_gpu_double_k_larger_than_the_tree(gpu):
    """k > n_points on a float64 tree: as the float32 entry (tests/test_gpu_parity.py::test_k_larger_than_the_tree):
    the n neighbours in order, the DBL_MAX sentinel in the last slot (search_visitor.hpp:95-110)."""
    rng = np.random.default_rng(44)
    pts, q = rng.random((7, 3)), rng.random((50, 3))
    t = pt.KdTree(pts, pt.Metric.L2Squared, 3, device=gpu)
    ref = oracle.Oracle(pts, 3, "port", dtype=np.float64)
    got, want = t.search_knn(q, 12), ref.search_knn(q, 7)
    assert got.shape == (50, 12) and same(got[:, :7], want["index"], want["distance"])
    assert np.all(got["distance"][:, 11] == np.finfo(np.float64).max)


@pytest.mark.gpu
def test_gpu_double_large_batch_in_pieces(gpu, monkeypatch):
    """More queries than one launch's stack block holds (deep tree: many slots per lane)."""
    monkeypatch.setenv("PTK_STACK64_MB", "64")
    rng = np.random.default_rng(5)
    pts = np.cumsum(rng.random((40_000, 3)) ** 8, axis=0)  # a drawn-out curve: a deep, skewed tree
    q = pts[rng.integers(0, len(pts), 600_000)] + rng.normal(0, 1e-3, (600_000, 3))
    t = pt.KdTree(pts, pt.Metric.L2Squared, 1, device=gpu)
    ref = oracle.Oracle(pts, 1, "port", dtype=np.float64)
    ref.set_threads(os.cpu_count() or 1)
    want = ref.search_knn(q, 1)
    got = t.search_knn(q, 1)
    assert t.info()["max_depth"] > 20
    assert same(got.reshape(-1, 1), want["index"], want["distance"])
