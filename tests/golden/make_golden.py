#!/usr/bin/env python3
"""Generates the committed golden vectors from the COMPILED REFERENCE.

Run in the authoring container only (needs ``oracle/_ref/libptk_ref.so``, i.e.
``/root/reference`` present at build time):

    python tests/golden/make_golden.py

Every expected output below is produced by the reference's own headers
(``oracle/ref_driver.cpp``: kd_tree ctor, search_knn, search_radius, search_box,
kd_tree::save) built with the canonical flags ``-O3 -ffp-contract=off``.  The
inputs come from the portable SplitMix generator in ``pico_tree_amd.datasets``
(recorded in the file too, so the fixtures do not depend on it staying
unchanged).  What is stored is data only -- inputs, expected outputs and
checksums -- never reference source.

Files
-----
``g_small_3d.npz``   4096 points / 4096 queries, 3-D, leaf 10: knn 1, knn 16,
                     radius (traversal order and sorted distances), approximate
                     knn, box search, the reference's ``kd_tree::save`` byte stream.
``g_small_2d.npz``   2048 / 2048, 2-D, leaf 8 (compile-time dim 2 in the reference).
``g_small_5d.npz``   2048 / 1024, 5-D, leaf 6 (run-time dim in the reference).
``g_ties_3d.npz``    lattice points with many equal distances and duplicates.
``micro.json``       the 10-point worked example of SURVEY.md appendix A.
``hashes.json``      SHA-256 checksums of (index, distance-bits) for BASELINE
                     config 1 (100 k / 100 k uniform, knn 1) and a 1 M / 250 k
                     slice of config 2's clouds.
"""

from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from pico_tree_amd import datasets as ds  # noqa: E402


def digest(data: bytes) -> str:
    return hashlib.sha256(data).hexdigest()


def make_set(path, pts, q, leaf, ks, radius, e, box_half):
    ref = oracle.Oracle(pts, leaf, "reference")
    data = {"points": pts, "queries": q, "max_leaf_size": np.int64(leaf),
            "radius": np.float32(radius), "e": np.float32(e),
            "save_stream": np.frombuffer(ref.save_bytes(), dtype=np.uint8)}
    for k in ks:
        data[f"knn{k}"] = ref.search_knn(q, k)
    data[f"aknn{ks[-1]}"] = ref.search_knn(q, ks[-1], e=e)
    off, flat = ref.search_radius(q, radius)
    data["radius_offsets"], data["radius_flat"] = off, flat
    off_s, flat_s = ref.search_radius(q, radius, sort=True)
    assert np.array_equal(off, off_s)
    data["radius_sorted_distance"] = flat_s["distance"].copy()
    off_a, flat_a = ref.search_radius(q, radius, e=e)
    data["aradius_offsets"], data["aradius_flat"] = off_a, flat_a
    nb = min(256, len(q))
    mins, maxs = q[:nb] - np.float32(box_half), q[:nb] + np.float32(box_half)
    boff, bflat = ref.search_box(mins, maxs)
    data["box_mins"], data["box_maxs"] = mins, maxs
    data["box_offsets"], data["box_flat"] = boff, bflat
    np.savez_compressed(path, **data)
    print(f"{os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB, "
          f"radius hits/query {off[-1] / len(q):.1f}, box hits/query {boff[-1] / nb:.1f}")


def main():
    if not oracle.have_reference():
        raise SystemExit("the compiled reference (oracle/_ref) is required")

    make_set(os.path.join(HERE, "g_small_3d.npz"),
             ds.uniform_cloud(4096, 3, seed=1), ds.uniform_cloud(4096, 3, seed=2),
             leaf=10, ks=(1, 16), radius=0.004, e=1.44, box_half=0.05)
    make_set(os.path.join(HERE, "g_small_2d.npz"),
             ds.uniform_cloud(2048, 2, seed=3), ds.uniform_cloud(2048, 2, seed=4),
             leaf=8, ks=(1, 7), radius=0.002, e=1.21, box_half=0.03)
    make_set(os.path.join(HERE, "g_small_5d.npz"),
             ds.uniform_cloud(2048, 5, seed=5), ds.uniform_cloud(1024, 5, seed=6),
             leaf=6, ks=(1, 5), radius=0.08, e=1.5, box_half=0.15)
    lattice = (np.round(ds.uniform_cloud(4096, 3, seed=7) * 8) / 8).astype(np.float32)
    lq = (np.round(ds.uniform_cloud(2048, 3, seed=8) * 16) / 16).astype(np.float32)
    make_set(os.path.join(HERE, "g_ties_3d.npz"), lattice, lq,
             leaf=10, ks=(1, 12), radius=0.05, e=1.3, box_half=0.1)

    # 10 points, leaf 2 -- re-derived here from the reference.
    pts = np.array([[1, 1], [2, 8], [3, 3], [9, 9], [4, 7], [8, 2], [6, 5], [7, 6], [2, 2], [9, 1]],
                   dtype=np.float32)
    ref = oracle.Oracle(pts, 2, "reference")
    q = np.array([[5, 5], [2.5, 2.5]], dtype=np.float32)
    off10, flat10 = ref.search_radius(q[:1], 10.0)
    off05, flat05 = ref.search_radius(q[1:], 0.5)
    micro = {
        "points": pts.tolist(), "max_leaf_size": 2, "queries": q.tolist(),
        "save_stream_hex": ref.save_bytes().hex(),
        "nn": [[int(r["index"]), float(r["distance"])] for r in ref.search_nn(q)],
        "knn3_q0": [[int(r["index"]), float(r["distance"])] for r in ref.search_knn(q[:1], 3)[0]],
        "radius10_q0": [[int(r["index"]), float(r["distance"])] for r in flat10],
        "radius0.5_q1_count": int(off05[-1]),
    }
    with open(os.path.join(HERE, "micro.json"), "w") as f:
        json.dump(micro, f, indent=1)
    print("micro.json:", micro["nn"], micro["knn3_q0"], micro["radius10_q0"], micro["radius0.5_q1_count"])

    hashes = {}
    p1, q1 = ds.uniform_cloud(100_000, 3, seed=1), ds.uniform_cloud(100_000, 3, seed=2)
    r = oracle.Oracle(p1, 10, "reference").search_knn(q1, 1)
    hashes["config1_uniform_100k_knn1"] = {
        "generator": "uniform_cloud(100000, 3, seed=1) / uniform_cloud(100000, 3, seed=2), leaf 10",
        "index_sha256": digest(r["index"].tobytes()), "distance_bits_sha256": digest(r["distance"].tobytes()),
        "index_sum": int(r["index"].astype(np.int64).sum())}
    for cloud in ("L", "U"):
        p2, q2 = ds.config2_clouds(cloud, 1_000_000, 250_000)
        ref2 = oracle.Oracle(p2, 10, "reference")
        for k in (1, 16):
            r = ref2.search_knn(q2, k)
            hashes[f"config2_{cloud}_1M_250k_knn{k}"] = {
                "generator": f"config2_clouds('{cloud}', 1000000, 250000), leaf 10",
                "index_sha256": digest(r["index"].tobytes()),
                "distance_bits_sha256": digest(r["distance"].tobytes()),
                "index_sum": int(r["index"].astype(np.int64).sum())}
    with open(os.path.join(HERE, "hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1)
    print("hashes.json written:", list(hashes))


if __name__ == "__main__":
    main()
