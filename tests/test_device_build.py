"""The top levels of ptk_tree_create_from_points built on the device (pico_tree_amd/csrc/ptk_build.hpp): the tree
must be the one the host builder makes -- nodes, permutation, outer bounds -- and therefore the reference's
(tests/test_oracle.py pins the host builder to the compiled reference node for node)."""
import numpy as np
import pytest

import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

pytestmark = pytest.mark.gpu


def _clouds():
    rng = np.random.default_rng(5)
    yield "uniform", ds.uniform_cloud(700_000, 3, 11)
    pts, _ = ds.config2_clouds("L", 1_200_000, 10)
    yield "lidar", pts
    sheets = ds.uniform_cloud(600_000, 3, 12)
    sheets[::3, 2] *= np.float32(1e-4)  # a dense sheet: lopsided splits, many levels above the threshold
    sheets[1::7] = sheets[::7][: len(sheets[1::7])]  # exact duplicates (ties on every plane they touch)
    yield "sheets+duplicates", sheets
    yield "2-d", np.ascontiguousarray(ds.uniform_cloud(500_000, 3, 13)[:, :2])
    yield "5-d", rng.random((400_000, 5), dtype=np.float32)
    quant = np.round(ds.uniform_cloud(600_000, 3, 14) / np.float32(0.01)) * np.float32(0.01)  # points ON the planes
    yield "quantised", quant.astype(np.float32)


@pytest.mark.parametrize("threads", ["3", "64"])
def test_device_built_tree_is_the_host_built_tree(gpu, monkeypatch, threads):
    monkeypatch.setenv("PTK_BUILD_THREADS", threads)
    for name, pts in _clouds():
        monkeypatch.setenv("PTK_DEVICE_BUILD", "0")
        host = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
        monkeypatch.setenv("PTK_DEVICE_BUILD", "1")
        dev = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
        a, b = host.flat(), dev.flat()
        for x, y in zip(a, b):
            assert np.asarray(x).tobytes() == np.asarray(y).tobytes(), name
        assert dev.info() == host.info(), name
        q = np.ascontiguousarray(pts[:: max(1, len(pts) // 2000)])
        if pts.shape[1] <= 3:
            assert dev.search_knn(q, 4).tobytes() == host.search_knn(q, 4).tobytes(), name


def test_planes_that_slide_in_the_top_levels(gpu, monkeypatch):
    """Most of the root box empty along its longest side: the first partitions leave one side empty and the builder
    slides the plane (std::nth_element): those ranges make a round trip to the host, the tree is the same."""
    pts = ds.uniform_cloud(400_000, 3, 21)
    pts[:, 0] *= np.float32(0.2)
    pts[0, 0] = np.float32(50.0)  # one far point: the box is long in x, everything else sits in its first fifth
    monkeypatch.setenv("PTK_DEVICE_BUILD", "0")
    host = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    monkeypatch.setenv("PTK_DEVICE_BUILD", "1")
    dev = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    for x, y in zip(host.flat(), dev.flat()):
        assert np.asarray(x).tobytes() == np.asarray(y).tobytes()
