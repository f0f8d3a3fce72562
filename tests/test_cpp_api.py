"""The header-only C++ host API (include/pico_tree) exercised from real C++ programs.

CPU tier
  * tests/cpp/host_api_main.cpp built WITHOUT linking libptk (-DPTK_TEST_HOST_ONLY): the
    per-query members (search_nn / search_knn / search_radius / search_box, custom visitor,
    save / load, run-time dimensions, other rules and metrics, double) against the oracle,
    bit for bit; the save stream against the oracle's restatement of kd_tree::save.
  * In the authoring container, the REFERENCE's own example sources are compiled in place
    against these headers (with a small stand-in for its pico_toolshed helpers) -- the
    acceptance check of SURVEY.md 8(b).  Skipped where /root/reference does not exist.
GPU tier
  * the same program linked against libptk.so: the batched members through the C ABI.
"""

from __future__ import annotations

import os
import subprocess

import numpy as np
import pytest

import oracle
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_api_main.cpp")
INC = os.path.join(ROOT, "include")
K, RADIUS = 7, np.float32(0.0009)


def _compile(out, host_only):
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wall", "-pthread", "-I" + INC, SRC, "-o", out]
    if host_only:
        cmd.insert(1, "-DPTK_TEST_HOST_ONLY")
    else:
        libdir = os.path.join(ROOT, "pico_tree_amd", "csrc")
        cmd += ["-L" + libdir, "-lptk", "-Wl,-rpath," + libdir]
    subprocess.check_call(cmd)


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    d = tmp_path_factory.mktemp("cppapi")
    pts = ds.uniform_cloud(30_000, 3, seed=91)
    q = ds.uniform_cloud(2_000, 3, seed=92)
    q[:50] = pts[:50]  # some queries sit exactly on tree points
    pts.tofile(os.path.join(d, "points.bin"))
    q.tofile(os.path.join(d, "queries.bin"))
    return str(d), pts, q


def _load(d, name, dtype):
    return np.fromfile(os.path.join(d, name), dtype=dtype)


def test_host_members_match_oracle(workdir):
    d, pts, q = workdir
    exe = os.path.join(d, "host_only")
    _compile(exe, host_only=True)
    subprocess.check_call([exe, "host", d])
    ref = oracle.Oracle(pts, 10, "port")
    assert _load(d, "nn.bin", pt.NEIGHBOR).tobytes() == ref.search_nn(q).tobytes()
    assert _load(d, "knn.bin", pt.NEIGHBOR).tobytes() == ref.search_knn(q, K).tobytes()
    assert _load(d, "aknn.bin", pt.NEIGHBOR).tobytes() == ref.search_knn(q, K, e=1.44).tobytes()
    off, flat = ref.search_radius(q, RADIUS)
    assert np.array_equal(_load(d, "radius_off.bin", np.uint64), off)
    assert _load(d, "radius_flat.bin", pt.NEIGHBOR).tobytes() == flat.tobytes()
    _, sflat = ref.search_radius(q, RADIUS, sort=True)
    assert np.array_equal(_load(d, "radius_sorted.bin", pt.NEIGHBOR)["distance"], sflat["distance"])
    lo, hi = q - np.float32(0.02), q + np.float32(0.02)
    boff, bflat = ref.search_box(lo, hi)
    assert np.array_equal(_load(d, "box_off.bin", np.uint64), boff)
    assert np.array_equal(_load(d, "box_flat.bin", np.int32), bflat)
    # the custom visitor saw exactly the points the reference traversal measures
    _, cnt = ref.search_nn(q, counters=True)
    assert np.array_equal(_load(d, "visits.bin", np.int32), cnt[:, 2].astype(np.int32))
    # byte-compatible tree file
    assert _load(d, "save.bin", np.uint8).tobytes() == ref.save_bytes()


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference not present")
def test_topological_metrics_match_the_compiled_reference(workdir):
    """metric_so2 / metric_se2_squared (reference metric.hpp:186-257, search_nearest_topological
    kd_tree_search.hpp:115-229) run on the host members of include/pico_tree; checked bit for bit
    against the reference's own headers (oracle/_ref, kd_tree<space, metric_so2|metric_se2_squared>)."""
    d, pts, q = workdir
    exe = os.path.join(d, "host_only")
    if not os.path.exists(os.path.join(d, "t_se2_knn.bin")):
        _compile(exe, host_only=True)
        subprocess.check_call([exe, "host", d])
    so2 = oracle.Oracle(np.ascontiguousarray(pts[:, :1]), 10, "reference", "SO2")
    se2 = oracle.Oracle(pts, 10, "reference", "SE2Squared")
    assert _load(d, "t_so2_knn.bin", pt.NEIGHBOR).tobytes() == so2.search_knn(np.ascontiguousarray(q[:, :1]), K).tobytes()
    assert _load(d, "t_se2_knn.bin", pt.NEIGHBOR).tobytes() == se2.search_knn(q, K).tobytes()
    assert _load(d, "t_se2_aknn.bin", pt.NEIGHBOR).tobytes() == se2.search_knn(q, K, e=1.44).tobytes()
    off, flat = se2.search_radius(q, RADIUS)
    assert off[-1] > 0 and np.array_equal(_load(d, "t_se2_radius_off.bin", np.uint64), off)
    assert _load(d, "t_se2_radius_flat.bin", pt.NEIGHBOR).tobytes() == flat.tobytes()
    lo, hi = q - np.float32(0.02), q + np.float32(0.02)
    through = (lo[:, 2] < 0) | (hi[:, 2] > 1)          # the angle's interval through the seam: min above max
    lo[:, 2] = np.where(lo[:, 2] < 0, lo[:, 2] + np.float32(1.0), lo[:, 2])
    hi[:, 2] = np.where(hi[:, 2] > 1, hi[:, 2] - np.float32(1.0), hi[:, 2])
    boff, bflat = se2.search_box(np.ascontiguousarray(lo), np.ascontiguousarray(hi))
    assert np.array_equal(_load(d, "t_se2_box_off.bin", np.uint64), boff)
    assert np.array_equal(_load(d, "t_se2_box_flat.bin", np.int32), bflat)
    assert through.sum() > 20 and np.diff(boff)[through].sum() > 0
    assert _load(d, "t_se2_save.bin", np.uint8).tobytes() == se2.save_bytes()
    # wrap-around really happens in this data: some neighbours are nearer through 0 ~ 1
    knn = _load(d, "t_so2_knn.bin", pt.NEIGHBOR).reshape(-1, K)
    direct = np.abs(pts[knn["index"], 0] - q[:, :1])
    assert (direct > 0.5).any()


REF_EXAMPLES = "/root/reference/examples/kd_tree"


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="reference sources not present")
@pytest.mark.parametrize("example", ["kd_tree_minimal", "kd_tree_creation", "kd_tree_custom_point_type",
                                     "kd_tree_custom_space_type", "kd_tree_custom_search_visitor",
                                     "kd_tree_dynamic_arrays", "kd_tree_save_and_load", "kd_tree_search",
                                     "kd_tree_custom_metric"])
def test_reference_examples_compile_unchanged(example, tmp_path):
    """Drop-in check: the reference's example programs, compiled where they lie, against
    include/pico_tree.  Only a stand-in for its pico_toolshed test helpers is ours."""
    exe = str(tmp_path / example)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + INC,
                           "-I" + os.path.join(ROOT, "tests", "cpp", "toolshed_shim"),
                           os.path.join(REF_EXAMPLES, example + ".cpp"), "-o", exe])
    subprocess.check_call([exe], stdout=subprocess.DEVNULL, cwd=str(tmp_path))


@pytest.mark.gpu
def test_batched_members_match_oracle(workdir, gpu):
    d, pts, q = workdir
    exe = os.path.join(d, "with_backend")
    _compile(exe, host_only=False)
    subprocess.check_call([exe, "batch", d])
    ref = oracle.Oracle(pts, 10, "port")
    assert _load(d, "b_nn.bin", pt.NEIGHBOR).tobytes() == ref.search_nn(q).tobytes()
    assert _load(d, "b_knn.bin", pt.NEIGHBOR).tobytes() == ref.search_knn(q, K).tobytes()
    assert _load(d, "b_aknn.bin", pt.NEIGHBOR).tobytes() == ref.search_knn(q, K, e=1.44).tobytes()
    off, flat = ref.search_radius(q, RADIUS)
    assert np.array_equal(_load(d, "b_radius_off.bin", np.uint64), off)
    assert _load(d, "b_radius_flat.bin", pt.NEIGHBOR).tobytes() == flat.tobytes()
    _, sflat = ref.search_radius(q, RADIUS, sort=True)
    assert np.array_equal(_load(d, "b_radius_sorted.bin", pt.NEIGHBOR)["distance"], sflat["distance"])
    boff, bflat = ref.search_box(q - np.float32(0.02), q + np.float32(0.02))
    assert np.array_equal(_load(d, "b_box_off.bin", np.uint64), boff)
    assert np.array_equal(_load(d, "b_box_flat.bin", np.int32), bflat)
    # other metrics: trees over the QUERY cloud, searched with itself
    l1, linf = oracle.Oracle(q, 10, "port", "L1"), oracle.Oracle(q, 10, "port", "LPInf")
    assert _load(d, "b_l1_knn.bin", pt.NEIGHBOR).tobytes() == l1.search_knn(q, K).tobytes()
    assert _load(d, "b_linf_knn.bin", pt.NEIGHBOR).tobytes() == linf.search_knn(q, K).tobytes()
    off, flat = l1.search_radius(q, 0.03)
    assert np.array_equal(_load(d, "b_l1_radius_off.bin", np.uint64), off)
    assert _load(d, "b_l1_radius_flat.bin", pt.NEIGHBOR).tobytes() == flat.tobytes()
    lninf = oracle.Oracle(q, 10, "port", "LNInf")
    assert _load(d, "b_lninf_knn.bin", pt.NEIGHBOR).tobytes() == lninf.search_knn(q, K).tobytes()
    if oracle.have_reference():  # the topological metric against the reference's own kd_tree<space, metric_se2_squared>
        se2 = oracle.Oracle(q, 10, "reference", "SE2Squared")
        assert _load(d, "b_se2_knn.bin", pt.NEIGHBOR).tobytes() == se2.search_knn(q, K).tobytes()
        off, flat = se2.search_radius(q, RADIUS)
        assert np.array_equal(_load(d, "b_se2_radius_off.bin", np.uint64), off)
        assert _load(d, "b_se2_radius_flat.bin", pt.NEIGHBOR).tobytes() == flat.tobytes()
    # double precision members (ptk_tree64_* / ptk_search64_*): the driver scales the clouds by 1.0000001 in double
    pd, qd = pts.astype(np.float64) * 1.0000001, q.astype(np.float64) * 1.0000001
    r64 = oracle.Oracle(pd, 10, "port", dtype=np.float64)
    want = r64.search_knn(qd, K)
    assert np.array_equal(_load(d, "d_knn_idx.bin", np.int32).reshape(-1, K), want["index"])
    assert _load(d, "d_knn_dist.bin", np.float64).tobytes() == np.ascontiguousarray(want["distance"]).tobytes()
    off, flat = r64.search_radius(qd, 0.0009)
    assert np.array_equal(_load(d, "d_radius_off.bin", np.uint64), off)
    assert np.array_equal(_load(d, "d_radius_idx.bin", np.int32), flat["index"])
    assert _load(d, "d_radius_dist.bin", np.float64).tobytes() == np.ascontiguousarray(flat["distance"]).tobytes()
    boff, bflat = r64.search_box(qd - 0.02, qd + 0.02)
    assert np.array_equal(_load(d, "d_box_off.bin", np.uint64), boff)
    assert np.array_equal(_load(d, "d_box_flat.bin", np.int32), bflat)
    if oracle.have_reference64():  # kd_tree<space of double points, metric_se2_squared> on the device
        sq = q.astype(np.float64)
        se2d = oracle.Oracle(sq, 10, "reference", "SE2Squared", dtype=np.float64)
        want = se2d.search_knn(sq, K)
        assert np.array_equal(_load(d, "d_se2_knn_idx.bin", np.int32).reshape(-1, K), want["index"])
        assert _load(d, "d_se2_knn_dist.bin", np.float64).tobytes() == np.ascontiguousarray(want["distance"]).tobytes()
        off, flat = se2d.search_radius(sq, 0.0009)
        assert off[-1] > 0 and np.array_equal(_load(d, "d_se2_radius_off.bin", np.uint64), off)
        assert np.array_equal(_load(d, "d_se2_radius_idx.bin", np.int32), flat["index"])
        assert _load(d, "d_se2_radius_dist.bin", np.float64).tobytes() == np.ascontiguousarray(flat["distance"]).tobytes()
        boff, bflat = se2d.search_box(sq - 0.02, sq + 0.02)
        assert np.array_equal(_load(d, "d_se2_box_off.bin", np.uint64), boff)
        assert np.array_equal(_load(d, "d_se2_box_flat.bin", np.int32), bflat)
