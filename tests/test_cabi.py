"""CPU tier: libptk.so loads, exports exactly what include/ptk.h declares, and behaves on a
machine without a GPU the way the boundary promises: structure queries work on a host-only
handle, every search fails LOUDLY (no CPU fallback), bad arguments are rejected."""

from __future__ import annotations

import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptk.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptk_[a-z_0-9]+)\s*\(", text)))


def test_header_is_plain_c():
    """The boundary must compile as C (no C++ types leak through)."""
    src = '#include "ptk.h"\nint main(void){ptk_tree_desc d; (void)d; return sizeof(ptk_neighbor)==8?0:1;}\n'
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    "-x", "c", "-", "-fsyntax-only"], input=src.encode(), check=True)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(pt.library_path())
    names = declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in ptk.h but not exported"
    assert set(pt.EXPORTED_SYMBOLS) == set(names), "Python binding table out of date"
    assert pt._load().ptk_version() == 101


def test_host_only_handle_and_loud_failures():
    pts = ds.uniform_cloud(5000, 3, seed=1)
    tree = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=pt.PTK_DEVICE_NONE)
    info = tree.info()
    assert info["n_points"] == 5000 and info["dim"] == 3 and info["device"] == pt.PTK_DEVICE_NONE
    assert info["n_nodes"] == 2 * info["n_leaves"] - 1 and info["max_leaf_count"] <= 10
    nodes, idx, rmin, rmax = tree.flat()
    assert sorted(idx.tolist()) == list(range(5000))
    assert np.array_equal(rmin, pts.min(0)) and np.array_equal(rmax, pts.max(0))
    q = ds.uniform_cloud(10, 3, seed=2)
    with pytest.raises(pt.PtkError) as e:
        tree.search_knn(q, 1)
    assert e.value.status == -3 and "no device replica" in str(e.value)
    with pytest.raises(pt.PtkError):
        tree.search_radius(q, 0.1)


def test_argument_validation():
    lib = pt._load()
    h = ctypes.c_void_p()
    pts = ds.uniform_cloud(100, 3, seed=1)
    assert lib.ptk_tree_create_from_points(None, 100, 3, 10, pt.PTK_DEVICE_NONE, ctypes.byref(h)) == -1
    assert lib.ptk_tree_create_from_points(pts.ctypes.data, 0, 3, 10, pt.PTK_DEVICE_NONE, ctypes.byref(h)) == -1
    assert lib.ptk_tree_create_from_points(pts.ctypes.data, 100, 0, 10, pt.PTK_DEVICE_NONE, ctypes.byref(h)) == -1
    assert lib.ptk_tree_create_from_points(pts.ctypes.data, 100, 3, 0, pt.PTK_DEVICE_NONE, ctypes.byref(h)) == -1
    assert b"positive" in lib.ptk_last_error()
    with pytest.raises(ValueError):
        pt.KdTree(pts.astype(np.float16))
    with pytest.raises(ValueError):
        pt.KdTree(pts[::2])
    with pytest.raises(TypeError):
        pt.KdTree(pts, "L1")
    t = pt.KdTree(pts, pt.Metric.L1, device=pt.PTK_DEVICE_NONE)
    assert lib.ptk_tree_set_metric(t._h, 4) == -1 and lib.ptk_tree_set_metric(t._h, -1) == -1
    assert lib.ptk_tree_set_metric(t._h, pt._PTK_METRIC[pt.Metric.LPInf]) == 0
    # F-ordered (sdim, npts) input is the same memory as C-ordered (npts, sdim): accepted
    t = pt.KdTree(np.asfortranarray(pts.T), pt.Metric.L2Squared, 10, device=pt.PTK_DEVICE_NONE)
    assert t.npts == 100 and t.sdim == 3 and t.metric(-2.0) == 4.0


def test_rejects_malformed_flat_trees():
    """ptk_tree_create validates the node stream instead of trusting it."""
    from ctypes import POINTER, Structure, c_float, c_int32, c_uint32, c_uint64, c_void_p

    class Desc(Structure):
        _fields_ = [("dim", c_uint32), ("n_points", c_uint64), ("points", c_void_p), ("n_nodes", c_uint64),
                    ("nodes", c_void_p), ("indices", c_void_p), ("root_min", c_void_p), ("root_max", c_void_p),
                    ("max_depth", c_uint32), ("device", c_int32)]

    lib = pt._load()
    pts = ds.uniform_cloud(64, 3, seed=3)
    good = pt.KdTree(pts, pt.Metric.L2Squared, 8, device=pt.PTK_DEVICE_NONE)
    nodes, idx, _, _ = good.flat()

    def create(n, i):
        d = Desc(3, 64, pts.ctypes.data, len(n), n.ctypes.data, i.ctypes.data, None, None, 0, pt.PTK_DEVICE_NONE)
        h = c_void_p()
        rc = lib.ptk_tree_create(ctypes.byref(d), ctypes.byref(h))
        if rc == 0:
            lib.ptk_tree_destroy(h)
        return rc

    assert create(nodes, idx) == 0
    bad = nodes.copy(); bad[0, 2] = 1                      # right child must lie after the left subtree
    assert create(bad, idx) == -1
    bad = nodes.copy(); bad[0, 3] = 7                      # split axis >= dim
    assert create(bad, idx) == -1
    assert create(nodes[:-1].copy(), idx) == -1            # truncated stream
    leaf = np.flatnonzero(nodes[:, 2] == 0xFFFFFFFF)[0]
    bad = nodes.copy(); bad[leaf, 1] = 1000                # leaf range past the end
    assert create(bad, idx) == -1
    # Leaves must tile [0, n) in stream order: the device layout places a leaf's points by a running
    # sum of the leaf sizes, so swapped or overlapping ranges would search the wrong points silently.
    leaves = np.flatnonzero(nodes[:, 2] == 0xFFFFFFFF)
    bad = nodes.copy(); bad[leaves[0], :2], bad[leaves[1], :2] = nodes[leaves[1], :2], nodes[leaves[0], :2]
    assert create(bad, idx) == -1
    assert b"previous leaf" in lib.ptk_last_error()
    bad = nodes.copy(); bad[leaves[-1], 1] -= 1            # the last leaf stops short of n
    assert create(bad, idx) == -1


def test_tree_stream_round_trip_on_a_host_only_handle():
    """ptk_tree_serialize writes the reference's kd_tree::save format (equal to the oracle's
    stream, which test_oracle.py pins against the compiled reference) and
    ptk_tree_create_from_stream reads it back to the same flat tree; bad streams fail loudly."""
    import ctypes
    from ctypes import byref, c_uint64, c_void_p

    import oracle
    from pico_tree_amd import datasets as ds

    lib = pt._load()
    pts = ds.uniform_cloud(3_000, 3, 4)
    h = c_void_p()
    assert lib.ptk_tree_create_from_points(pts.ctypes.data, len(pts), 3, 9, pt.PTK_DEVICE_NONE, byref(h)) == 0
    size = c_uint64()
    assert lib.ptk_tree_serialize(h, None, 0, byref(size)) == 0
    buf = ctypes.create_string_buffer(size.value)
    assert lib.ptk_tree_serialize(h, buf, size.value - 1, byref(size)) == -1  # too small
    assert lib.ptk_tree_serialize(h, buf, size.value, byref(size)) == 0
    assert buf.raw[:size.value] == oracle.Oracle(pts, 9, "port").save_bytes()
    h2 = c_void_p()
    assert lib.ptk_tree_create_from_stream(pts.ctypes.data, len(pts), 3, buf, size.value, pt.PTK_DEVICE_NONE,
                                           byref(h2)) == 0
    size2 = c_uint64()
    buf2 = ctypes.create_string_buffer(size.value)
    assert lib.ptk_tree_serialize(h2, buf2, size.value, byref(size2)) == 0 and buf2.raw == buf.raw
    h3 = c_void_p()
    assert lib.ptk_tree_create_from_stream(pts.ctypes.data, len(pts) - 1, 3, buf, size.value, pt.PTK_DEVICE_NONE,
                                           byref(h3)) == -1  # wrong point count
    assert lib.ptk_tree_create_from_stream(pts.ctypes.data, len(pts), 3, buf, 100, pt.PTK_DEVICE_NONE,
                                           byref(h3)) == -1  # truncated
    for x in (h, h2):
        lib.ptk_tree_destroy(x)


def test_threaded_build_gives_the_identical_tree(monkeypatch):
    """PTK_BUILD_THREADS only changes who builds which subtree: same nodes, same permutation."""
    from ctypes import byref, c_void_p

    from pico_tree_amd import datasets as ds

    lib = pt._load()
    pts = np.concatenate([ds.lidar_cloud(150_000, 1), ds.lidar_cloud(50_000, 1)])  # with exact duplicates
    flats = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("PTK_BUILD_THREADS", threads)
        h = c_void_p()
        assert lib.ptk_tree_create_from_points(pts.ctypes.data, len(pts), 3, 10, pt.PTK_DEVICE_NONE, byref(h)) == 0
        inf = pt._Info()
        assert lib.ptk_tree_get_info(h, byref(inf)) == 0
        nodes = np.empty((inf.n_nodes, 4), dtype=np.uint32)
        idx = np.empty(len(pts), dtype=np.int32)
        assert lib.ptk_tree_get_flat(h, nodes.ctypes.data, idx.ctypes.data, None, None) == 0
        flats.append((nodes, idx, inf.max_depth, inf.n_leaves))
        lib.ptk_tree_destroy(h)
    for other in flats[1:]:
        assert np.array_equal(other[0], flats[0][0]) and np.array_equal(other[1], flats[0][1])
        assert other[2:] == flats[0][2:]


def test_degenerate_point_set_is_refused_not_crashed():
    """Thousands of identical points with a small leaf size make the sliding midpoint peel one
    point per level: the reference recurses until its stack overflows; the build here stops at
    8192 levels with PTK_ERR_UNSUPPORTED."""
    pts = np.full((60_000, 1), 0.5e-6, dtype=np.float32)
    with pytest.raises(pt.PtkError, match="deeper than 8192"):
        pt.KdTree(pts, pt.Metric.L2Squared, 5, device=pt.PTK_DEVICE_NONE)
    t = pt.KdTree(np.full((6_000, 3), 1.5, dtype=np.float32), pt.Metric.L2Squared, 2, device=pt.PTK_DEVICE_NONE)
    assert len(t.flat()[0]) == 11_997  # deep (2 999 levels) but built


def test_multi_device_entry_points_fail_loudly_without_a_device():
    """ptk_multi_*: argument checks, and no silent single-device or CPU stand-in."""
    import torch
    lib = pt._load()
    pts = ds.uniform_cloud(100, 3, seed=5)
    h = ctypes.c_void_p()
    dev = np.zeros(1, dtype=np.int32)
    assert lib.ptk_multi_create_from_points(pts.ctypes.data, 100, 3, 10, None, 0, ctypes.byref(h)) == -1
    if not torch.cuda.is_available():
        rc = lib.ptk_multi_create_from_points(pts.ctypes.data, 100, 3, 10, dev.ctypes.data, 1, ctypes.byref(h))
        assert rc == -3 and not h.value  # PTK_ERR_DEVICE
    assert lib.ptk_multi_device_count(None) == 0
    assert lib.ptk_multi_search_knn(None, pts.ctypes.data, 1, 1, ctypes.c_float(1.0), None) == -1


def test_input_file_formats_round_trip(tmp_path):
    """CPU tier: the reference's write_bin / read_bin (format_bin.hpp:9-32) and xvecs readers
    (format_xvecs.hpp:41-67) as pico_tree_amd.datasets reads them."""
    from pico_tree_amd import datasets as ds
    pts = ds.uniform_cloud(1000, 3, 5)
    ds.write_bin(str(tmp_path / "scan.bin"), pts)
    assert ds.read_bin(str(tmp_path / "scan.bin")).tobytes() == pts.tobytes()
    with open(tmp_path / "scan.bin", "ab") as f:
        f.write(b"\x00" * 7)  # a torn tail is ignored (file_size / sizeof(point))
    assert ds.load_points(str(tmp_path / "scan.bin")).tobytes() == pts.tobytes()
    rows = ds.sift_like_cloud(50, 128, seed=3)
    ds.write_xvecs(str(tmp_path / "base.fvecs"), rows)
    raw = np.fromfile(tmp_path / "base.fvecs", dtype=np.uint8)
    assert len(raw) == 50 * (4 + 4 * 128) and raw[:4].view("<i4")[0] == 128
    assert ds.read_xvecs(str(tmp_path / "base.fvecs")).tobytes() == rows.tobytes()
    ds.write_xvecs(str(tmp_path / "base.bvecs"), rows)
    b = ds.read_xvecs(str(tmp_path / "base.bvecs"))
    assert b.dtype == np.uint8 and np.array_equal(b.astype(np.float32), rows)  # SIFT bytes
    assert ds.load_points(str(tmp_path / "base.bvecs")).dtype == np.float32
    with pytest.raises(ValueError):
        ds.read_xvecs(str(tmp_path / "scan.bin"))


def test_key_bits_follow_the_splits_of_the_tree():
    """The Morton key of a batch spends its bits where the tree divides space: evenly on a cube, (almost) none
    on the axis a flat cloud does not extend along; 24 bits for a big batch, 16 below 2^20 queries, 15 at
    most per axis; fewer axes share all of them."""
    cube = ds.uniform_cloud(40_000, 3, 1)
    t = pt.KdTree(cube, pt.Metric.L2Squared, 10, device=pt.PTK_DEVICE_NONE)
    assert t.key_bits(5_000_000) == (8, 8, 8)
    b = t.key_bits(1000)
    assert sum(b) == 16 and max(b) - min(b) <= 1
    flat = cube.copy()
    flat[:, 2] *= np.float32(1e-4)  # a sheet: the sliding midpoint rule splits the longest side, never z
    t = pt.KdTree(flat, pt.Metric.L2Squared, 10, device=pt.PTK_DEVICE_NONE)
    bx, by, bz = t.key_bits(5_000_000)
    assert bz == 0 and bx + by == 24 and abs(bx - by) <= 1
    long = cube.copy()
    long[:, 0] *= np.float32(4096.0)  # a rod: x gets the most a single axis may have
    t = pt.KdTree(long, pt.Metric.L2Squared, 10, device=pt.PTK_DEVICE_NONE)
    bx, by, bz = t.key_bits(5_000_000)
    assert bx >= 10 and bx <= 15 and bx + by + bz == 24
    t = pt.KdTree(np.ascontiguousarray(cube[:, :2]), pt.Metric.L2Squared, 10, device=pt.PTK_DEVICE_NONE)
    assert t.key_bits(5_000_000) == (12, 12, 0)
    t = pt.KdTree(cube[:1], pt.Metric.L2Squared, 10, device=pt.PTK_DEVICE_NONE)  # one leaf: nothing to go by
    assert t.key_bits(5_000_000) == (8, 8, 8)


def test_the_reference_module_name_is_importable():
    """``import pico_tree`` (the reference binding's module name) resolves to this package's classes."""
    import pico_tree

    assert pico_tree.KdTree is pt.KdTree and pico_tree.DArray is pt.DArray and pico_tree.Metric is pt.Metric
    assert pico_tree.load_kd_tree is pt.load_kd_tree and pico_tree.save_kd_tree is pt.save_kd_tree


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference not present")
@pytest.mark.parametrize("metric,dim", [("SO2", 1), ("SE2Squared", 3)])
def test_topological_tree_stream_is_the_reference_stream(tmp_path, metric, dim):
    """kd_tree<space, metric_so2 | metric_se2_squared>::save writes four bounds per branch
    (kd_tree_branch_double, internal/kd_tree_node.hpp:52-67): a tree built here serialises to the same bytes, and a
    saved one loads straight into a handle with its outer bounds (no device needed for either)."""
    pts = ds.uniform_cloud(6_000, dim, 17)
    tree = pt.KdTree(pts, pt.Metric[metric], 10, device=pt.PTK_DEVICE_NONE)
    ref = oracle.Oracle(pts, 10, "reference", metric)
    stream = tree._serialize()
    assert stream == ref.save_bytes()
    path = str(tmp_path / "topological.pkd")
    pt.save_kd_tree(tree, path)
    again = pt.load_kd_tree(pts, path, device=pt.PTK_DEVICE_NONE)
    assert again.metric_string == metric and again._serialize() == stream
    for a, b in zip(tree.flat(), again.flat()):
        assert np.array_equal(a, b)
    plain = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=pt.PTK_DEVICE_NONE)
    assert len(plain._serialize()) < len(stream)  # two bounds per branch fewer


def test_knn_cap_follows_the_batch(monkeypatch):
    """ptk_debug_knn_cap: the far children a query of a k > 1 search may enter before a wavefront takes it over.  It
    grows with the batch (a capped launch ends with the lanes at their cap, so a small batch wants a low one), between a
    floor and a top that follow k; approximate searches, tiny batches and k outside 2 .. 56 run uncapped; the list of
    hand-overs always has room for the first 24 576 (or all) queries."""
    lib = pt._load()

    def cap(nq, k, e=1.0):
        c, entries = ctypes.c_uint32(), ctypes.c_uint64()
        assert lib.ptk_debug_knn_cap(nq, k, np.float32(e), ctypes.byref(c), ctypes.byref(entries)) == 0
        return c.value, entries.value

    monkeypatch.delenv("PTK_TEST_KNOBS", raising=False)
    sizes = [32, 256, 1_000, 20_000, 150_000, 600_000, 900_000, 2_400_000, 7_200_863, 50_000_000]
    for k, floor, top in ((2, 8, 256), (4, 8, 256), (8, 12, 320), (16, 16, 448), (32, 32, 512), (56, 64, 768)):
        caps = [cap(nq, k)[0] for nq in sizes]
        assert caps == sorted(caps) and caps[0] == floor and caps[-1] == top, (k, caps)
    assert cap(7_200_863, 16)[0] == 448 and cap(900_000, 16)[0] == 56 and cap(900_000, 4)[0] == 24
    assert cap(31, 16) == (0, 0) and cap(32, 16) == (16, 32) and cap(10_000, 16, e=1.5) == (0, 0) and cap(10_000, 1) == (0, 0) and cap(10_000, 57) == (0, 0)
    for nq in sizes:
        entries = cap(nq, 16)[1]
        assert min(nq, 24_576) <= entries <= max(nq // 48, 24_576)
    pt.set_test_knobs(knn_cap=77)
    assert cap(5_000, 16)[0] == 77 and cap(7_200_863, 4)[0] == 77
    pt.set_test_knobs(knn_cap=0)
    assert cap(7_200_863, 16) == (0, 0)
    pt.set_test_knobs(knn_cap=None, knn_cap_min_nq=1)
    assert cap(3, 16)[0] == 16
    assert lib.ptk_debug_knn_cap(10, 16, np.float32(1.0), None, None) == -1


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference not present")
@pytest.mark.parametrize("metric,dim", [("SO2", 1), ("SE2Squared", 3)])
def test_host_loop_box_search_of_a_topological_tree(metric, dim):
    """ptk_host_search_box on a tree over a topological space (what serves a box search the device refuses: a tree
    deeper than its stack) is the reference's search_box with a metric_box_map query (box.hpp:300-376): an interval
    of the circle axis may run through the seam 0 ~ 1 (min above max), and every axis takes the four-bound
    intersection tests.  A lattice cloud: thousands of coincident angles, the deep tree of the case the fuzzer found."""
    rng = np.random.default_rng(9)
    pts = np.ascontiguousarray(np.round(rng.random((8_000, dim)) * 6) / 6, dtype=np.float32)
    pts[:, -1] = np.where(pts[:, -1] >= 1, np.float32(0.0), pts[:, -1])
    q = rng.random((400, dim)).astype(np.float32)
    q[:100, -1] = (rng.random(100) * 1e-3).astype(np.float32)
    lo, hi = q - np.float32(0.03), q + np.float32(0.03)
    through = (lo[:, -1] < 0) | (hi[:, -1] > 1)
    lo[:, -1] = np.where(lo[:, -1] < 0, lo[:, -1] + np.float32(1.0), lo[:, -1])
    hi[:, -1] = np.where(hi[:, -1] > 1, hi[:, -1] - np.float32(1.0), hi[:, -1])
    lo, hi = np.ascontiguousarray(lo), np.ascontiguousarray(hi)
    tree = pt.KdTree(pts, pt.Metric[metric], 4, device=pt.PTK_DEVICE_NONE)
    ref = oracle.Oracle(pts, 4, "reference", metric)
    lib = pt._load()
    offsets = np.zeros(len(q) + 1, dtype=np.uint64)
    rows = ctypes.c_void_p()
    assert lib.ptk_host_search_box(tree._h, pts.ctypes.data, lo.ctypes.data, hi.ctypes.data, len(q),
                                   offsets.ctypes.data, ctypes.byref(rows)) == 0
    flat = np.ctypeslib.as_array(ctypes.cast(rows, ctypes.POINTER(ctypes.c_int32)), shape=(int(offsets[-1]),)).copy()
    lib.ptk_free(rows)
    boff, bflat = ref.search_box(lo, hi)
    assert np.array_equal(offsets, boff) and np.array_equal(flat, bflat)
    assert through.sum() >= 100 and np.diff(boff)[through].sum() > 0


@pytest.mark.parametrize("metric", ["L2Squared", "L1", "LPInf"])
def test_host_loop_entry_points_equal_the_oracle(metric):
    """ptk_host_search_* (what the wrappers call when a DEVICE search answers PTK_ERR_UNSUPPORTED): the reference's
    batch loop over the handle's flat tree.  Runs without a device -- and is the only search that does: ptk_search_*
    of the same handle still fail loudly (test_host_only_handle_and_loud_failures)."""
    pts = ds.uniform_cloud(6000, 3, seed=5)
    q = ds.uniform_cloud(700, 3, seed=6)
    tree = pt.KdTree(pts, pt.Metric[metric], 7, device=pt.PTK_DEVICE_NONE)
    ref = oracle.Oracle(pts, 7, "port", metric)
    lib = pt._load()
    for k, e in ((1, 1.0), (5, 1.0), (9, 1.7)):
        out = np.empty((len(q), k), dtype=pt.NEIGHBOR)
        assert lib.ptk_host_search_knn(tree._h, pts.ctypes.data, q.ctypes.data, len(q), k, np.float32(e),
                                       out.ctypes.data) == 0
        assert out.tobytes() == ref.search_knn(q, k, e=None if e == 1.0 else e).tobytes()
    radius = np.float32(0.004 if metric == "L2Squared" else 0.06)
    for e, sort in ((1.0, 0), (2.0, 0)):
        offsets = np.zeros(len(q) + 1, dtype=np.uint64)
        rows = ctypes.c_void_p()
        assert lib.ptk_host_search_radius(tree._h, pts.ctypes.data, q.ctypes.data, len(q), radius, np.float32(e), sort,
                                          offsets.ctypes.data, ctypes.byref(rows)) == 0
        flat = np.ctypeslib.as_array(ctypes.cast(rows, ctypes.POINTER(ctypes.c_uint8)),
                                     shape=(int(offsets[-1]) * 8,)).copy().view(pt.NEIGHBOR)
        lib.ptk_free(rows)
        off, want = ref.search_radius(q, radius, e=None if e == 1.0 else e)
        assert np.array_equal(offsets, off) and flat.tobytes() == want.tobytes() and off[-1] > 0
    lo, hi = np.ascontiguousarray(q - np.float32(0.03)), np.ascontiguousarray(q + np.float32(0.03))
    offsets = np.zeros(len(q) + 1, dtype=np.uint64)
    rows = ctypes.c_void_p()
    assert lib.ptk_host_search_box(tree._h, pts.ctypes.data, lo.ctypes.data, hi.ctypes.data, len(q),
                                   offsets.ctypes.data, ctypes.byref(rows)) == 0
    flat = np.ctypeslib.as_array(ctypes.cast(rows, ctypes.POINTER(ctypes.c_int32)), shape=(int(offsets[-1]),)).copy()
    lib.ptk_free(rows)
    boff, bflat = ref.search_box(lo, hi)
    assert np.array_equal(offsets, boff) and np.array_equal(flat, bflat)
