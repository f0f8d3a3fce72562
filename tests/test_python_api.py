"""The reference's Python unit tests (/root/reference/test/pyco_tree/kd_tree_test.py)
restated against ``pico_tree_amd.KdTree``: same calls, same assertions (float32 here; the float64
cases of the reference's tests are in tests/test_f64.py).  Anything this build does not run on the
device is a loud error, never a silent CPU fallback.
"""

from __future__ import annotations

import copy

import numpy as np
import pytest

import pico_tree_amd as pt

pytestmark = pytest.mark.gpu

A = [[2, 1], [4, 3], [8, 7]]


def test_creation_kd_tree(gpu):  # kd_tree_test.py:11-42
    a = np.array(A, dtype=np.float32, order="C")
    t = pt.KdTree(a, pt.Metric.L2Squared, 10, device=gpu)
    assert (a.shape[0], a.shape[1]) == (t.npts, t.sdim)
    assert a.dtype == t.dtype_scalar
    with pytest.raises(ValueError):  # non-contiguous
        pt.KdTree(a[::2], pt.Metric.L2Squared, 10, device=gpu)
    f = np.array(A, dtype=np.float32, order="F")  # column major: (sdim, npts)
    t = pt.KdTree(f, pt.Metric.L2Squared, 10, device=gpu)
    assert (f.shape[1], f.shape[0]) == (t.npts, t.sdim)
    with pytest.raises(ValueError):  # must have two dimensions
        pt.KdTree(np.array([[[2, 1]], [[4, 3]], [[8, 7]]], dtype=np.float32), pt.Metric.L2Squared, 10, device=gpu)
    d = np.array(A, dtype=np.float64, order="C")  # kd_tree_test.py:13-18: the scalar dtype follows the input
    t = pt.KdTree(d, pt.Metric.L2Squared, 10, device=gpu)
    assert d.dtype == t.dtype_scalar and (t.npts, t.sdim) == d.shape
    with pytest.raises(ValueError):  # queries of another dtype than the tree's
        t.search_knn(a, 1)
    with pytest.raises(ValueError):  # neither float32 nor float64
        pt.KdTree(np.array(A, dtype=np.int64), pt.Metric.L2Squared, 10, device=gpu)


def test_metric(gpu):  # kd_tree_test.py:44-51
    a = np.array(A, dtype=np.float32)
    t = pt.KdTree(a, pt.Metric.L2Squared, 10, device=gpu)
    assert t.metric(-2.0) == 4
    t = pt.KdTree(a, pt.Metric.L1, 10, device=gpu)
    assert t.metric(-2.0) == 2
    assert "metric=L1" in repr(t)
    t = pt.KdTree(a, pt.Metric.LPInf, 10, device=gpu)
    assert t.metric(-2.0) == 2


@pytest.mark.parametrize("e", [None, 1.0])  # kd_tree_test.py:53-89 (exact and approximate)
def test_search_knn(gpu, e):
    a = np.array(A, dtype=np.float32)
    t = pt.KdTree(a, pt.Metric.L2Squared, 10, device=gpu)
    k = 2
    args = (a, k) if e is None else (a, k, e)
    nns = t.search_knn(*args)
    assert nns.shape == (3, k)
    for i in range(len(nns)):
        assert nns[i][0][0] == i
        assert nns[i][0][1] == pytest.approx(0)
    data = copy.deepcopy(nns.ctypes.data)  # the memory is re-used
    t.search_knn(*args, nns)
    assert nns.ctypes.data == data


@pytest.mark.parametrize("e", [None, 1.0])  # kd_tree_test.py:91-153
def test_search_radius(gpu, e):
    a = np.array(A, dtype=np.float32)
    t = pt.KdTree(a, pt.Metric.L2Squared, 10, device=gpu)
    radius = t.metric(2.5)
    args = (a, radius) if e is None else (a, radius, e)
    nns = t.search_radius(*args)
    assert len(nns) == 3
    assert nns.dtype == t.dtype_neighbor
    assert nns
    for i, n in enumerate(nns):
        assert len(n) == 1
        assert n[0][0] == i
        assert n[0][1] == pytest.approx(0)
    for i in range(len(nns)):  # DArray is a sequence
        assert nns[i][0][0] == i

    def addresses(rows):
        return [copy.deepcopy(x.ctypes.data) if len(x) else 0 for x in rows]

    datas = addresses(nns)
    t.search_radius(*args, nns)
    assert addresses(nns) == datas


def test_creation_darray(gpu):  # kd_tree_test.py:203-229
    a = np.array(A, dtype=np.float32)
    t = pt.KdTree(a, pt.Metric.L2Squared, 10, device=gpu)
    d = pt.DArray(t.dtype_neighbor)
    assert d.dtype == t.dtype_neighbor and not d
    d = pt.DArray(dtype=t.dtype_neighbor)
    assert d.dtype == t.dtype_neighbor and not d
    for spec in (np.int32, np.dtype(np.int32)):
        d = pt.DArray(spec)
        assert d.dtype == t.dtype_index and not d


def test_darray_slicing_and_negative_index(gpu):  # the sequence checks of kd_tree_test.py:193-201
    pts = pt.datasets.uniform_cloud(500, 3, 3)
    t = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    nns = t.search_radius(pts[:4], 0.01)
    sub = nns[0:4:2]
    assert len(sub) == 2
    assert [len(x) for x in sub] == [len(nns[0]), len(nns[2])]
    assert len(nns[-1]) == len(nns[3])
    assert sub[1].tobytes() == nns[2].tobytes()


def test_column_major_queries_give_transposed_output(gpu):  # _pyco_tree/kd_tree.hpp:362-378
    pts = pt.datasets.uniform_cloud(300, 2, 9)
    t = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    q = pts[:50]
    row = t.search_knn(q, 3)
    col = t.search_knn(np.asfortranarray(q.T), 3)  # (sdim, npts) F-order = same memory
    assert row.shape == (50, 3) and col.shape == (3, 50)
    assert col.reshape(-1).tobytes() == row.reshape(-1).tobytes()


def test_search_box(gpu):  # kd_tree_test.py:155-201
    a = np.array(A, dtype=np.float32)
    t = pt.KdTree(a, pt.Metric.L2Squared, 10, device=gpu)
    boxes = np.array([[0, 0], [3, 3], [2, 2], [3, 3], [0, 0], [9, 9], [6, 6], [9, 9]], dtype=np.float32)
    nns = t.search_box(boxes)
    assert len(nns) == 4
    assert nns.dtype == t.dtype_index
    assert nns

    def addresses(rows):
        return [copy.deepcopy(x.ctypes.data) if len(x) else 0 for x in rows]

    datas = addresses(nns)
    t.search_box(boxes, nns)
    assert addresses(nns) == datas
    assert [len(n) for n in nns] == [1, 0, 3, 1]
    sub = nns[0:4:2]
    assert len(sub) == 2 and [len(n) for n in sub] == [1, 3]
    assert len(sub[-1]) == 3
    with pytest.raises(ValueError):
        t.search_box(boxes[:3])


def test_file_io(gpu, tmp_path):  # kd_tree_test.py:231-248
    a = np.array(A, dtype=np.float32, order="C")
    t1 = pt.KdTree(a, pt.Metric.L2Squared, 10, device=gpu)
    filename = str(tmp_path / "tree.bin")
    pt.save_kd_tree(t1, filename)
    t2 = pt.load_kd_tree(a, filename, device=gpu)
    k = 2
    assert repr(t1) == repr(t2)
    assert t1.dtype_scalar == t2.dtype_scalar
    assert np.array_equal(t1.search_knn(a, k), t2.search_knn(a, k))
    with open(filename, "r+b") as f:
        f.write(b"XXXX")
    with pytest.raises(RuntimeError):
        pt.load_kd_tree(a, filename, device=gpu)


def test_file_io_is_byte_compatible_with_the_reference(gpu, tmp_path):
    """The tree part of the file is the reference's own kd_tree::save stream (pinned here through
    the oracle's save_bytes, which tests/test_oracle.py pins against the compiled reference), and a
    stream produced by the reference side loads and answers identically."""
    import oracle
    pts = pt.datasets.uniform_cloud(5_000, 3, 8)
    q = pt.datasets.uniform_cloud(1_000, 3, 9)
    t1 = pt.KdTree(pts, pt.Metric.L2Squared, 7, device=gpu)
    ref = oracle.Oracle(pts, 7, "port")
    assert t1._serialize() == ref.save_bytes()
    filename = str(tmp_path / "ref.bin")
    with open(filename, "wb") as f:  # what the reference's save_kd_tree writes
        f.write(b"\x89PKD" + np.uint32(1).tobytes() + np.uint64(9).tobytes() + b"L2Squared" + ref.save_bytes())
    t2 = pt.load_kd_tree(pts, filename, device=gpu)
    assert t2.search_knn(q, 5).tobytes() == ref.search_knn(q, 5).tobytes()
    got = t2.search_radius(q, 0.002)
    off, flat = ref.search_radius(q, 0.002)
    assert np.array_equal(got.offsets, off) and got.flat.tobytes() == flat.tobytes()



def test_import_pico_tree_runs_a_script_written_for_the_reference(gpu, tmp_path, monkeypatch):
    """``import pico_tree as pt`` (examples/python/kd_tree.py:5): the calls of the reference's Python example, made
    through the alias package -- default device, no device argument, as a script written for the reference makes them."""
    import pico_tree as ref_name

    assert ref_name.KdTree is pt.KdTree and ref_name.Metric is pt.Metric and ref_name.DArray is pt.DArray
    assert set(["DArray", "Metric", "KdTree", "load_kd_tree", "save_kd_tree"]) <= set(ref_name.__all__)
    monkeypatch.chdir(tmp_path)
    p = np.array(A, dtype=np.float32)
    for metric in (ref_name.Metric.L1, ref_name.Metric.LPInf, ref_name.Metric.L2Squared):  # creation, kd_tree.py:14-33
        t = ref_name.KdTree(p, metric, 10)
        assert str(t) and (t.npts, t.sdim) == (3, 2)
    t = ref_name.KdTree(p, ref_name.Metric.L2Squared, 1)
    assert t.metric(-2.0) == 4.0
    knns = t.search_knn(p, 1)                                # :42-49
    assert [int(i) for i in knns["index"].reshape(-1)] == [0, 1, 2] and not knns["distance"].any()
    t.search_knn(p, 2, knns)
    assert knns.shape == (3, 2) and [int(i) for i in knns[:, 1]["index"]] == [1, 0, 1]
    ratio = t.metric(1.0 + 0.75)                             # :57-66
    knns = t.search_knn(p, 2, ratio)
    t.search_knn(p, 2, ratio, knns)
    assert np.allclose(knns[:, 1]["distance"] * ratio, [8.0, 8.0, 32.0])
    rnns = t.search_radius(p, t.metric(2.5))                 # :73-82
    assert [len(r) for r in rnns] == [1, 1, 1]
    t.search_radius(p, 25.0, rnns)
    assert [len(r) for r in rnns] == [2, 2, 1]
    boxes = np.array([[0, 0], [3, 3], [2, 2], [3, 3], [0, 0], [9, 9], [6, 6], [9, 9]], dtype=np.float32)  # :91-103
    bnns = t.search_box(boxes)
    t.search_box(boxes, bnns)
    assert [list(b) for b in bnns] == [[0], [], [0, 1, 2], [2]]
    assert len(bnns) == 4 and list(bnns[0]) == [0] and list(bnns[-2]) == [0, 1, 2]   # :111-117
    assert [list(b) for b in bnns[0:4:2]] == [[0], [0, 1, 2]]
    a = np.array(A, dtype=np.float64, order="C")             # file io, :130-143
    t1 = ref_name.KdTree(a, ref_name.Metric.L2Squared, 10)
    ref_name.save_kd_tree(t1, "tree.bin")
    t2 = ref_name.load_kd_tree(a, "tree.bin")
    assert repr(t1) == repr(t2) and t2.search_knn(a, 1)["index"].reshape(-1).tolist() == [0, 1, 2]
