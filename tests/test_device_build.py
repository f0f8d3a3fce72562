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


@pytest.mark.parametrize("case", ["far-max", "far-min", "far-max-quantised", "far-min-duplicates"])
def test_planes_that_slide_in_the_top_levels(gpu, monkeypatch, case):
    """Most of the root box empty along its longest side: the first partitions leave one side empty and the builder
    slides the plane (std::nth_element).  The first rounds of libstdc++'s introselect run on the device (the Hoare
    partitions of ptk_build.hpp: pairing of left and right stops by rank), the rest of the same call on the host:
    the permutation -- and with it the tree -- must be the host's to the last index.  Far point beyond the maximum
    (nth = last - 1: the ranges shrink from the left) and below the minimum (nth = first + 1: from the right);
    coordinates on a grid (thousands of elements EQUAL to the pivot: both kinds of stop at once) and exact duplicates."""
    pts = ds.uniform_cloud(1_500_000, 3, 21)
    pts[:, 0] *= np.float32(0.2)
    if "quantised" in case:
        pts[:, 0] = np.round(pts[:, 0] / np.float32(0.002)) * np.float32(0.002)
    if "duplicates" in case:
        pts[1::5] = pts[::5][: len(pts[1::5])]
    pts[0, 0] = np.float32(50.0) if "max" in case else np.float32(-50.0)  # the box is long in x, the cloud at one end
    monkeypatch.setenv("PTK_DEVICE_BUILD", "0")
    host = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    monkeypatch.setenv("PTK_DEVICE_BUILD", "1")
    dev = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    for x, y in zip(host.flat(), dev.flat()):
        assert np.asarray(x).tobytes() == np.asarray(y).tobytes()
    pt.set_test_knobs(device_slide_off=1)  # (the round trip to the host for the whole range: the same tree)
    dev2 = pt.KdTree(pts, pt.Metric.L2Squared, 10, device=gpu)
    for x, y in zip(host.flat(), dev2.flat()):
        assert np.asarray(x).tobytes() == np.asarray(y).tobytes()
