"""CPU tier: the product's REAL kernel source, compiled for the host against a HIP
stand-in (tests/cpp/hip_stub) and run lane by lane, against the oracle and the
golden vectors.  This checks the traversal logic the GPU executes -- 8-byte
record stack with LDS + scratch halves, undo records, push-time pruning, parent
re-read on far descents, stable k-list insertion, the two radius passes, the
permuted (Morton) launch -- on a machine without a GPU.  The `-m gpu` tests
repeat the comparison on the device.
"""

from __future__ import annotations

import os

import numpy as np
import pytest

import oracle
import pico_tree_amd as pt
from pico_tree_amd import datasets as ds
from tests.emu import EmulatedTree
from tests.test_oracle import check_against_golden, load


class _Adaptor:
    """Gives EmulatedTree the call shape check_against_golden expects."""

    def __init__(self, emu):
        self.emu = emu

    def search_knn(self, q, k, e=None):
        return self.emu.search_knn(q, k, e=e)

    def search_radius(self, q, radius, sort=False, e=None):
        return self.emu.search_radius(q, radius, sort=sort, e=e)


@pytest.mark.parametrize("name", ["g_small_3d", "g_small_2d", "g_ties_3d"])
def test_emulated_kernels_reproduce_golden_vectors(name):
    g = load(name)
    emu = EmulatedTree(g["points"], int(g["max_leaf_size"]))
    check_against_golden(_Adaptor(emu), g)


def _cases():
    yield "uniform", ds.uniform_cloud(20_000, 3, 1), ds.uniform_cloud(3_000, 3, 2), 10, 0.002
    yield "dim2", ds.uniform_cloud(5_000, 2, 1), ds.uniform_cloud(2_000, 2, 2), 6, 0.001
    yield "dim1", ds.uniform_cloud(3_000, 1, 1), ds.uniform_cloud(1_000, 1, 2), 4, 1e-5
    p = (np.round(ds.uniform_cloud(20_000, 3, 5) * 8) / 8).astype(np.float32)
    yield "ties", p, (np.round(ds.uniform_cloud(3_000, 3, 6) * 16) / 16).astype(np.float32), 10, 0.05
    yield "self", p, p[:3_000].copy(), 3, 0.02
    yield "lidar", ds.lidar_cloud(30_000, 1), ds.lidar_cloud(3_000, 2, pose=(3.0, 1.5)), 10, 1.0
    yield "root-is-leaf", ds.uniform_cloud(7, 3, 1), ds.uniform_cloud(100, 3, 2), 10, 0.5
    yield "leaf1", ds.uniform_cloud(300, 3, 1), ds.uniform_cloud(300, 3, 2), 1, 0.05


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_emulated_kernels_equal_oracle(case):
    _, pts, q, leaf, radius = case
    emu = EmulatedTree(pts, leaf)
    ref = oracle.Oracle(pts, leaf, "port")
    perm, _ = emu.morton_permutation(q)
    assert sorted(perm.tolist()) == list(range(len(q)))
    for k in (1, 4, 16):
        if k > len(pts):
            continue
        want = ref.search_knn(q, k)
        for small_stack in (False, True):       # True: records spill to the scratch half
            for list_in_lds in (False, True, 2):  # k-list in the output row / in LDS / in registers
                for p in (None, perm):          # identity vs Morton launch order
                    got = emu.search_knn(q, k, perm=p, small_stack=small_stack, list_in_lds=list_in_lds)
                    assert got.tobytes() == want.tobytes()
        assert emu.search_knn(q, k, e=1.3).tobytes() == ref.search_knn(q, k, e=1.3).tobytes()
    for e in (None, 1.5):
        off, flat = ref.search_radius(q, radius, e=e)
        goff, gflat = emu.search_radius(q, radius, e=e, perm=perm)
        assert np.array_equal(goff, off) and gflat.tobytes() == flat.tobytes()
        _, sflat = ref.search_radius(q, radius, sort=True, e=e)
        _, gs = emu.search_radius(q, radius, sort=True, e=e)
        assert np.array_equal(gs["distance"], sflat["distance"])


@pytest.mark.parametrize("dim,n,nq,radius", [(4, 6_000, 1_500, 0.02), (5, 6_000, 1_500, 0.05), (16, 4_000, 600, 1.0),
                                             (100, 1_500, 150, 13.0)])
def test_emulated_any_dimension_kernels_equal_oracle(dim, n, nq, radius):
    """dim > 3: the any-dimension kernels (query / offset vectors in LDS)."""
    pts, q = ds.uniform_cloud(n, dim, 1), ds.uniform_cloud(nq, dim, 2)
    emu = EmulatedTree(pts, 8)
    ref = oracle.Oracle(pts, 8, "port")
    perm = np.random.default_rng(dim).permutation(nq).astype(np.uint32)  # any launch order, same rows
    for k in (1, 7, 20):
        want = ref.search_knn(q, k)
        for small_stack in (False, True):
            for list_in_lds in (False, True, 2):  # k-list in the output row / in LDS / in registers
                for p in (None, perm):
                    got = emu.search_knn(q, k, perm=p, small_stack=small_stack, list_in_lds=list_in_lds)
                    assert got.tobytes() == want.tobytes()
    assert emu.search_knn(q, 4, e=1.25).tobytes() == ref.search_knn(q, 4, e=1.25).tobytes()
    off, flat = ref.search_radius(q, radius)
    goff, gflat = emu.search_radius(q, radius)
    assert off[-1] > 0 and np.array_equal(goff, off) and gflat.tobytes() == flat.tobytes()


def test_block_mapping_is_a_permutation_in_runs():
    """xcd_runs(): every tile exactly once whatever the grid size, and in a complete group of 8 x 16 blocks the
    blocks that land on one XCD (block index mod 8) take 16 consecutive tiles."""
    import ctypes
    from tests.emu import _lib
    lib = _lib()
    lib.emu_xcd_runs.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
    lib.emu_xcd_runs.restype = None
    for nb in (1, 7, 8, 127, 128, 129, 255, 256, 1000, 112514):
        tiles = np.full(nb, 0xFFFFFFFF, dtype=np.uint32)
        lib.emu_xcd_runs(nb, tiles.ctypes.data)
        assert np.array_equal(np.sort(tiles), np.arange(nb, dtype=np.uint32)), nb
        for g in range(nb // 128):
            grp = tiles[g * 128:(g + 1) * 128]
            for x in range(8):
                assert np.array_equal(grp[x::8], g * 128 + x * 16 + np.arange(16)), (nb, g, x)
        assert np.array_equal(tiles[nb // 128 * 128:], np.arange(nb // 128 * 128, nb))  # the incomplete group


def test_morton_keys_follow_the_curve():
    pts = ds.uniform_cloud(5_000, 3, 3)
    emu = EmulatedTree(pts, 10)
    q = ds.uniform_cloud(4_000, 3, 4)
    jump = lambda o: float(np.linalg.norm(np.diff(q[o], axis=0), axis=1).mean())  # noqa: E731
    for bits, total in (((8, 8, 8), 24), ((10, 10, 10), 30), ((11, 9, 4), 24), ((15, 0, 9), 24), ((1, 1, 1), 3)):
        perm, keys = emu.morton_permutation(q, bits)
        assert keys.max() < (1 << total)
        assert sorted(perm.tolist()) == list(range(len(q)))
        if min(bits) >= 8:  # (even bits suit this cube) spatial neighbours end up together: mean jump length
            assert jump(perm) < 0.25 * jump(np.arange(len(q)))
    # Uneven bits interleave from the top: with (2, 1, 0) the key is x1 y0 x0 (x's first bit, then y's and x's).
    lo, hi = pts.min(0), pts.max(0)
    probe = lo + (hi - lo) * np.array([[0.80, 0.75, 0.5], [0.30, 0.75, 0.9], [0.55, 0.25, 0.1]], dtype=np.float32)
    _, keys = emu.morton_permutation(probe, (2, 1, 0))
    assert keys.tolist() == [0b111, 0b011, 0b100]


@pytest.mark.parametrize("nq,bits,tile", [(4_000, (6, 5, 5), 256), (4_096, (8, 8, 8), 512), (777, (3, 2, 1), 64),
                                          (64, (8, 8, 0), 64), (1, (8, 8, 8), 256), (9_001, (11, 10, 3), 1024),
                                          (9_001, (11, 10, 3), 0), (4_096, (6, 5, 5), 0), (4_097, (3, 2, 1), 0),
                                          (13_000, (8, 8, 0), 0), (100, (8, 8, 8), 0)])
def test_own_radix_sort_is_the_stable_sort_of_the_keys(nq, bits, tile):
    """ptk_sort.hpp (histogram / scan / scatter per 8-bit pass) against numpy's stable argsort of the same keys,
    incl. a ragged last tile, a last pass of fewer than 8 bits, heavy duplicates (6 key bits for 777 items).  tile = 0:
    the passes with blocks of four wavefronts on tiles of 4 096 items (256 fibers per block: ballots per wavefront,
    barriers per block)."""
    pts = ds.lidar_cloud(3_000, seed=5)
    emu = EmulatedTree(pts, 10)
    q = ds.lidar_cloud(nq, seed=6, pose=(1.0, 0.5))
    perm, keys = emu.radix_sorted_permutation(q, bits, tile)
    _, want_keys = emu.morton_permutation(q, bits)
    assert np.array_equal(keys, want_keys)
    assert np.array_equal(perm, np.argsort(keys, kind="stable").astype(np.uint32))


@pytest.mark.parametrize("case", [c for c in _cases() if c[0] in ("uniform", "ties", "self", "lidar", "root-is-leaf",
                                                                  "dim2", "leaf1")],
                         ids=lambda c: c[0])
def test_emulated_two_phase_knn1_equals_oracle(case):
    """The two-phase k = 1 search without the cap (what an approximate search runs): the wave-uniform-prefix
    phase 1 that packs the records itself, full key order, phase 2 to the end (3), and that form with one-point
    leaf batches, 4-slot rings and three narrow tiers of 1, 4 and 16 lanes per wavefront (4)."""
    _, pts, q, leaf, _ = case
    q = q[:1500]
    emu = EmulatedTree(pts, leaf)
    ref = oracle.Oracle(pts, leaf, "port")
    perm, _ = emu.morton_permutation(q)
    want = ref.search_knn(q, 1)
    for variant in (3, 4):
        for p in (None, perm):
            got, _ = emu.two_phase_knn1(q, perm=p, variant=variant)
            assert got.tobytes() == want.tobytes()
    got, _ = emu.two_phase_knn1(q, e=1.4, perm=perm, variant=3)
    assert got.tobytes() == ref.search_knn(q, 1, e=1.4).tobytes()


@pytest.mark.parametrize("case", [c for c in _cases() if c[0] in ("uniform", "self", "lidar", "dim2", "leaf1")],
                         ids=lambda c: c[0])
def test_emulated_two_phase_knn1_under_metric_l1_equals_oracle(case):
    """The two-phase k = 1 search instantiated over metric_l1 (r06: ptk_search_knn_device takes it for L1 trees without
    piles): uncapped (3), one-point leaf batches and narrow tiers (4), capped at 2 far children with the cooperative
    search, its certificate and the redo pass (5), cap 1 with 8 lanes per query (7), and the ranked classes straight
    from phase 1 to the cooperative search (9) -- every form bit-identical to the reference's kd_tree<space, metric_l1>."""
    _, pts, q, leaf, _ = case
    q = q[:1500]
    emu = EmulatedTree(pts, leaf, pt.Metric.L1)
    ref = oracle.Oracle(pts, leaf, "port", "L1")
    perm, _ = emu.morton_permutation(q)
    want = ref.search_knn(q, 1)
    for variant in (3, 4, 5, 7, 9):
        for p in (None, perm):
            got, _ = emu.two_phase_knn1(q, perm=p, variant=variant)
            assert got.tobytes() == want.tobytes(), variant
    got, _ = emu.two_phase_knn1(q, e=1.4, perm=perm, variant=3)
    assert got.tobytes() == ref.search_knn(q, 1, e=1.4).tobytes()


def _pile_cloud(n, dim, grid, seed, heavy=0):
    """Coordinates snapped to a grid: piles of coincident points (and `heavy` copies of one point on top)."""
    p = np.round(ds.uniform_cloud(n, dim, seed) / grid) * grid
    if heavy:
        p[:heavy] = p[0]
    return np.ascontiguousarray(p, dtype=np.float32)


@pytest.mark.parametrize("dim,grid,shift", [(3, 0.25, 0.0), (3, 0.25, 0.3), (3, 0.5, 0.5), (2, 0.125, 0.3), (1, 0.05, 0.5)])
def test_emulated_k1_search_on_the_view_without_the_piles(dim, grid, shift):
    """Trees of coincident points (ptk_piles.hpp): the k = 1 search on the view in which every pile is a leaf of one
    point, then the pass that gives a pile's row the point of it the reference visits first -- queries on the grid
    (on top of the piles), moved off it (every pile at a distance, several at the same one when shift = 0.5), exact
    and approximate, against the oracle; the full tree of the same handle gives the same rows the long way."""
    pts = _pile_cloud(6_000, dim, grid, 11, heavy=300)
    q = np.ascontiguousarray(np.round(ds.uniform_cloud(1_200, dim, 12) / grid) * grid + np.float32(shift * grid), dtype=np.float32)
    ref = oracle.Oracle(pts, 10, "port")
    want = ref.search_knn(q, 1)
    full = EmulatedTree(pts, 10)
    depth_full = full.host.info()["max_depth"]
    assert full.two_phase_knn1(q[:300], variant=5)[0].tobytes() == want[:300].tobytes()  # (the long way: depth > 300)
    emu = EmulatedTree(pts, 10)
    n_piles = emu.use_pile_view()
    assert n_piles > 10 and depth_full > 300
    perm, _ = emu.morton_permutation(q)
    for variant, pm in ((5, None), (5, perm), (3, perm), (9, None)):
        got, _ = emu.two_phase_knn1(q, perm=pm, variant=variant)
        assert got.tobytes() == want.tobytes(), (variant, pm is None)
    got, _ = emu.two_phase_knn1(q, e=1.3, perm=perm, variant=3)
    assert got.tobytes() == ref.search_knn(q, 1, e=1.3).tobytes()
    # a tree of points in general position has no pile: nothing to switch to
    assert EmulatedTree(ds.uniform_cloud(2_000, 3, 5), 10).use_pile_view() == 0


def _blind_disc_queries(n):
    """Queries inside the empty disc under the scanner of cloud L: the reference's depth-first search
    of such a query visits a long chain of leaves (the expensive queries of BASELINE config 2)."""
    u = ds.raw_uniform24(11, 2 * n).reshape(n, 2)
    r = 12.0 * np.sqrt(u[:, 0])
    a = 2.0 * np.pi * u[:, 1]
    return np.ascontiguousarray(np.stack([r * np.cos(a), r * np.sin(a), np.zeros(n)], axis=1), dtype=np.float32)


@pytest.mark.parametrize("case", [c for c in _cases() if c[0] in ("uniform", "ties", "self", "lidar", "root-is-leaf",
                                                                  "dim2", "dim1", "leaf1")],
                         ids=lambda c: c[0])
def test_emulated_capped_phase2_and_cooperative_search_equal_oracle(case):
    """Phase 2 with a cap on the far children a query may enter, the cooperative search (8 / 16 / 32 / 64
    lanes per query, fibers) for what is left and the redo pass for what that cannot certify: exact
    ties (lattice clouds, queries that are tree points) must come back through the redo list and
    still equal the reference."""
    name, pts, q, leaf, _ = case
    q = q[:1200]
    if name == "lidar":
        q = np.concatenate([q[:600], _blind_disc_queries(600)])
    emu = EmulatedTree(pts, leaf)
    ref = oracle.Oracle(pts, leaf, "port")
    perm, _ = emu.morton_permutation(q)
    want = ref.search_knn(q, 1)
    stats = {}
    for variant in (5, 6, 7, 8, 9):  # (9: the ranked classes straight from phase 1 to the cooperative search)
        # (the lattice of `ties` redoes most of its hand-overs lane by lane: the generated order once is enough there)
        for p in ((None, perm) if name != "ties" or variant == 5 else (perm,)):
            got, _ = emu.two_phase_knn1(q, perm=p, variant=variant)
            assert got.tobytes() == want.tobytes(), (name, variant)
        stats[variant] = emu.last_coop()
    if name in ("uniform", "lidar", "ties"):
        assert stats[5][0] > 0 and stats[6][0] >= stats[5][0]   # the cap does hand queries over
    if name in ("uniform", "lidar"):
        assert stats[5][1] <= stats[5][0] // 50                 # generic data: (almost) nothing to redo
    if name == "ties":                                          # exact ties are resolved by depth-first order
        assert stats[5][1] < stats[5][0] // 2
    if name == "self":                                          # a best of 0 ends the search in phase 1 already
        assert stats[5][0] == 0 and stats[5][1] == 0


@pytest.mark.parametrize("case", [c for c in _cases() if c[0] in ("uniform", "ties", "lidar", "root-is-leaf", "dim2",
                                                                  "leaf1")], ids=lambda c: c[0])
def test_emulated_capped_knn_and_its_cooperative_search_equal_oracle(case):
    """k > 1: the general kernel capped at a few far children per query, knn_coop_kernel (one wavefront of fibers per
    query handed over, every lane a k-list of its own, merged at the end) and the reference search of what that
    cannot certify -- equal distances within or at the edge of the k nearest (lattice clouds) must come back through
    the redo list -- equal the reference for every k-list size compiled."""
    name, pts, q, leaf, _ = case
    q = q[:100] if name == "ties" else q[:200]
    if name == "lidar":
        q = np.concatenate([q[:100], _blind_disc_queries(100)])
    emu = EmulatedTree(pts, leaf)
    ref = oracle.Oracle(pts, leaf, "port")
    perm, _ = emu.morton_permutation(q)
    seen = {}
    for k in ((2, 7, 16, 20, 40) if name == "uniform" else ((3, 16, 64) if name == "lidar" else (3, 16))):  # (K = 4 .. 64 slots compiled)
        kk = min(k, len(pts))
        want = ref.search_knn(q, kk)
        for cap, p, small in ((2, None, False), (1, perm, True)):
            got, heavy, redo = emu.search_knn_capped(q, kk, cap, perm=p, pool_small=small)
            assert got.tobytes() == want.tobytes(), (name, k, cap)
            seen[(k, cap)] = (heavy, redo)
    # more queries at their cap than the hand-over list holds: the rest finish in their lanes (Handover::full_keeps)
    kk = min(7, len(pts))
    got, heavy, redo = emu.search_knn_capped(q, kk, 1, max_heavy=3)
    assert got.tobytes() == ref.search_knn(q, kk).tobytes(), (name, "max_heavy")
    if name in ("uniform", "lidar"):
        assert heavy > 3 and redo <= 3  # (`heavy` counts every query that asked for an entry)
    if name in ("uniform", "lidar", "ties"):
        assert seen[(16, 2)][0] > 0                                   # the cap does hand queries over
    if name in ("uniform", "lidar"):
        assert seen[(16, 2)][1] <= max(2, seen[(16, 2)][0] // 20)     # generic data: (almost) nothing to redo
    if name == "ties":                        # equal distances everywhere: second sweeps, crowded ones redone, still exact
        assert seen[(16, 2)][1] > 0 and emu.last_tie_sweeps > 0


@pytest.mark.parametrize("name", ["uniform", "lidar", "ties"])
def test_emulated_capped_knn_under_metric_l1_equals_oracle(name):
    """The capped k > 1 search and its cooperative finish instantiated over metric_l1 (r06: launch_knn_reg takes it for
    L1 trees): hand-overs, second sweeps on the lattice cloud, redo -- rows bit-identical to kd_tree<space, metric_l1>."""
    pts, q, leaf = {"uniform": (ds.uniform_cloud(20_000, 3, 1), ds.uniform_cloud(300, 3, 2), 10),
                    "lidar": (ds.lidar_cloud(30_000, 1), ds.lidar_cloud(300, 2, pose=(3.0, 1.5)), 10),
                    "ties": ((np.round(ds.uniform_cloud(20_000, 3, 5) * 8) / 8).astype(np.float32),
                             (np.round(ds.uniform_cloud(100, 3, 6) * 16) / 16).astype(np.float32), 10)}[name]
    emu = EmulatedTree(pts, leaf, pt.Metric.L1)
    ref = oracle.Oracle(pts, leaf, "port", "L1")
    handed = swept = 0
    for k in (3, 16, 40):
        want = ref.search_knn(q, k)
        for cap in (1, 2):
            got, heavy, _ = emu.search_knn_capped(q, k, cap, pool_small=(cap == 1))
            assert got.tobytes() == want.tobytes(), (name, k, cap)
            handed += heavy
            swept += emu.last_tie_sweeps
    assert handed > 0 and (name != "ties" or swept > 0)


def test_emulated_cooperative_knn_on_a_line_of_points_with_drifting_box_distances():
    """The case the fuzz soak of r05 failed on (profiles/r05_notes.txt item 24): 60 000 points on a line in the plane,
    leaves of one point, queries off the line -- thousands of points at nearly the same distance, a tree a hundred levels
    deep whose incrementally updated box distances drift above the distances of the points inside.  The reference
    leaves such a point out; the second sweep of the cooperative search used to let it in (its box distances were
    below the k-th distance, but above its OWN): now such a query goes to the reference search.  Every list size."""
    # the draws of case 760, seed 802 of tools/fuzz_parity.py (one_case), in their order
    rng = np.random.default_rng([802, 760])
    dim = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 7]))
    n = int(rng.choice([1, 2, 9, 100, 3000, 20000, 60000]))
    nq = int(rng.choice([1, 63, 64, 65, 1000, 5000]))
    leaf = int(rng.choice([1, 2, 5, 10, 16, 24]))
    kind = str(rng.choice(["uniform", "clustered", "lattice", "duplicates", "line", "plane"]))
    scale = float(rng.choice([1.0, 1.0, 1e-6, 1e6, 37.5]))
    shift = float(rng.choice([0.0, 0.0, -0.5, 100.0]))
    rng.choice(["L2Squared", "L2Squared", "L1", "LPInf"])
    assert (dim, n, nq, leaf, kind, scale, shift) == (2, 60000, 64, 1, "line", 37.5, 0.0)
    pts = (((rng.random((n, 1)) * rng.random((1, dim)) + 0.25) + shift) * scale).astype(np.float32)
    assert rng.random() < 0.5
    q = (((rng.random((nq, 1)) * rng.random((1, dim)) + 0.25) + shift) * scale).astype(np.float32)[[15, 18, 26]]
    emu = EmulatedTree(pts, 1)
    ref = oracle.Oracle(pts, 1, "port")
    sweeps = 0
    for k in (16, 33):  # (rows 15 and 18 were wrong at k = 16, row 26 at k = 32 / 33 before the fix)
        got, heavy, redo = emu.search_knn_capped(q, k, 8)
        assert got.tobytes() == ref.search_knn(q, k).tobytes(), k
        sweeps += emu.last_tie_sweeps
        assert heavy > 0
    assert sweeps > 0  # equal distances at the edge of the list: second sweeps did run


@pytest.mark.parametrize("name", ["scan", "scan-lost", "uniform-leaf1", "big-leaves", "ties", "approximate"])
def test_emulated_capped_radius_search_and_its_cooperative_finish_equal_oracle(name):
    """The radius search with its long queries finished by a wavefront each (ptk_kernels_coopr.hpp): the list pass capped at a
    few far children per query, the cooperative count of what it handed over (any order; leaf entries keyed by their
    place in the reference's depth-first order, sorted), the recount from the root of what that could not finish (pool
    and spill full, more than 512 leaves with hits, the entry block exhausted); then the replay of the lanes' lists,
    the cooperative replay of the handed-over tails, and the ordinary fill kernel for what was lost.  Offsets and rows
    byte-equal to the reference's traversal-order rows in every regime."""
    kw, e = {}, None
    if name.startswith("scan"):
        pts = ds.lidar_cloud(120_000, seed=1, unit_scale=2.5)
        q = ds.lidar_cloud(150, seed=2, pose=(3.0, 1.5), unit_scale=2.5)
        leaf, radius, caps = 10, np.float32(1.0), (1, 8)
        if name == "scan-lost":  # a pool of 64 subtrees and 8 spill slots; room for 3 000 entries in all
            q, radius, caps, kw = q[:60], np.float32(4.0), (2,), {"pool_small": True, "entry_cap": 3000, "max_heavy": 50}
    elif name == "uniform-leaf1":  # more than 512 leaves with hits per query: recounted and refilled by one lane
        pts, q, leaf, radius, caps = ds.uniform_cloud(30_000, 3, 11), ds.uniform_cloud(40, 3, 12), 1, np.float32(0.05), (4,)
    elif name == "big-leaves":  # leaves of up to 40 points are listed in pieces of 32 (the key's piece bits order them)
        pts, q, leaf, radius, caps = ds.uniform_cloud(30_000, 3, 11), ds.uniform_cloud(200, 3, 12), 40, np.float32(0.01), (1, 4)
    elif name == "ties":
        pts = (np.round(ds.uniform_cloud(20_000, 3, 5) * 8) / 8).astype(np.float32)
        q = (np.round(ds.uniform_cloud(100, 3, 6) * 16) / 16).astype(np.float32)
        leaf, radius, caps = 10, np.float32(0.03), (1, 4)
    else:
        pts, q, leaf, radius, caps, e = ds.uniform_cloud(30_000, 3, 11), ds.uniform_cloud(40, 3, 12), 2, np.float32(0.02), (4,), 2.0
    emu = EmulatedTree(pts, leaf)
    ref = oracle.Oracle(pts, leaf, "port")
    want_off, want = ref.search_radius(q, radius, e=e)
    perm, _ = emu.morton_permutation(q)
    seen = {}
    for cap in caps:
        for p in (None, perm):
            off, rows, stats = emu.search_radius_lists_capped(q, radius, cap, e=e, perm=p, **kw)
            assert np.array_equal(off, want_off) and rows.tobytes() == want.tobytes(), (name, cap)
            seen[cap] = stats
    assert seen[caps[0]]["handed_over"] > 0
    if name in ("scan-lost", "uniform-leaf1"):
        assert seen[caps[0]]["recounted"] > 0 and seen[caps[0]]["refilled"] >= seen[caps[0]]["recounted"]
    else:
        assert seen[caps[0]]["recounted"] == 0


def test_emulated_direct_cooperative_search_parks_subtrees_in_hbm():
    """Variant 9 = the small-batch form: the ranked classes go straight from phase 1 to the cooperative search.  Its
    emulated launch has a pool of 12 subtrees per group, so the queries of the scanner's blind disc (hundreds of
    leaves each) must park subtrees in the HBM spill (and a few overflow even that and come back through the redo
    list): results equal the reference either way."""
    pts = ds.lidar_cloud(120_000, seed=1, unit_scale=20.0)
    q = (_blind_disc_queries(1500) * 20.0).astype(np.float32)
    emu = EmulatedTree(pts, 10)
    ref = oracle.Oracle(pts, 10, "port")
    got, _ = emu.two_phase_knn1(q, variant=9)
    assert got.tobytes() == ref.search_knn(q, 1).tobytes()
    assert emu.last_spilled() > 0 and emu.last_coop()[0] > 0


@pytest.mark.parametrize("case", [c for c in _cases() if c[0] in ("uniform", "dim2", "dim1", "ties", "lidar", "root-is-leaf",
                                                                  "leaf1")], ids=lambda c: c[0])
def test_emulated_box_search_equals_oracle(case):
    """search_box: running node box, wholesale-reported subtrees, closed bounds, traversal order."""
    _, pts, q, leaf, radius = case
    emu = EmulatedTree(pts, leaf)
    ref = oracle.Oracle(pts, leaf, "port")
    rng = np.random.default_rng(3)
    span = (pts.max(0) - pts.min(0)).astype(np.float32)
    h = (rng.uniform(0.01, 0.2, size=q.shape) * span).astype(np.float32)
    mins, maxs = (q - h).astype(np.float32), (q + h).astype(np.float32)
    mins[::20] = pts.min(0)  # boxes touching the root bounds exactly (closed interval)
    maxs[::20] = pts.max(0)
    maxs[1::20] = mins[1::20]
    off, flat = ref.search_box(mins, maxs)
    goff, gflat = emu.search_box(mins, maxs)
    assert np.array_equal(goff, off) and np.array_equal(gflat, flat)
    assert off[-1] > 0


@pytest.mark.parametrize("dim,leaf", [(4, 7), (6, 10), (17, 3)])
def test_emulated_box_search_any_dimension(dim, leaf):
    """box_nd_kernel (dim > 3): the same walk with the four per-lane vectors staged in LDS."""
    pts = ds.uniform_cloud(12_000, dim, 51)
    if dim == 4:
        pts = (np.round(pts * 6) / 6).astype(np.float32)  # lattice: points ON box faces
    q = ds.uniform_cloud(600, dim, 52)
    emu = EmulatedTree(pts, leaf)
    ref = oracle.Oracle(pts, leaf, "port")
    rng = np.random.default_rng(4)
    h = (rng.uniform(0.2, 0.6, size=q.shape)).astype(np.float32)
    mins, maxs = (q - h).astype(np.float32), (q + h).astype(np.float32)
    mins[::20] = pts.min(0)
    maxs[::20] = pts.max(0)
    maxs[1::20] = mins[1::20]
    off, flat = ref.search_box(mins, maxs)
    goff, gflat = emu.search_box(mins, maxs)
    assert np.array_equal(goff, off) and np.array_equal(gflat, flat)
    assert off[-1] > len(pts)


@pytest.mark.parametrize("metric", ["L1", "LPInf", "LNInf"])
@pytest.mark.parametrize("case", ["uniform3", "ties3", "dim2", "dim1", "dim5"])
def test_emulated_kernels_other_metrics(case, metric):
    """metric_l1 / metric_lpinf swapped into the generic kernels (the launches the backend makes
    for ptk_tree_set_metric != L2 squared), against the oracle under the same metric."""
    import pico_tree_amd as pt
    if case == "uniform3":
        pts, q, leaf = ds.uniform_cloud(20_000, 3, 31), ds.uniform_cloud(2_000, 3, 32), 10
    elif case == "ties3":
        pts = (np.round(ds.uniform_cloud(20_000, 3, 33) * 8) / 8).astype(np.float32)
        q, leaf = (np.round(ds.uniform_cloud(2_000, 3, 34) * 16) / 16).astype(np.float32), 10
    elif case == "dim2":
        pts, q, leaf = ds.uniform_cloud(8_000, 2, 35), ds.uniform_cloud(1_500, 2, 36), 5
    elif case == "dim1":
        pts, q, leaf = ds.uniform_cloud(3_000, 1, 39), ds.uniform_cloud(600, 1, 40), 4
    else:
        pts, q, leaf = ds.uniform_cloud(8_000, 5, 37), ds.uniform_cloud(800, 5, 38), 8
    emu = EmulatedTree(pts, leaf, pt.Metric[metric])
    ref = oracle.Oracle(pts, leaf, "port", metric)
    perm = emu.morton_permutation(q)[0] if pts.shape[1] <= 3 else None
    for k, small in ((1, False), (1, True), (4, True), (12, False), (40, False)):
        want = ref.search_knn(q, k)
        assert emu.search_knn(q, k, small_stack=small).tobytes() == want.tobytes(), (k, small)
        if perm is not None:
            assert emu.search_knn(q, k, perm=perm, small_stack=small).tobytes() == want.tobytes(), (k, small)
    assert emu.search_knn(q, 5, e=1.4).tobytes() == ref.search_knn(q, 5, e=1.4).tobytes()
    radius = 0.04 * float(np.ptp(pts, axis=0).max()) * (2.0 if pts.shape[1] > 3 else 1.0)
    if metric == "LNInf":  # the smallest coordinate difference: nearly every point is "near"
        radius *= 0.01
    for kw in ({}, {"e": 1.5}):
        a, b = emu.search_radius(q, radius, **kw), ref.search_radius(q, radius, **kw)
        assert np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes()
    a, b = emu.search_radius(q, radius, sort=True), ref.search_radius(q, radius, sort=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1]["distance"], b[1]["distance"])


def _rows_of_overfull_logs(offsets, perm, hits_per_log):
    """Rows of the wavefronts (64 consecutive entries of the launch order) that found more hits than
    `hits_per_log`: what a capture without a dynamic pool has to hand to the fill kernel."""
    counts = np.diff(offsets.astype(np.int64))
    order = np.arange(len(counts)) if perm is None else perm.astype(np.int64)
    rows = 0
    for lo in range(0, len(order), 64):
        tile = order[lo:lo + 64]
        if counts[tile].sum() > hits_per_log:
            rows += len(tile)
    return rows


def _emulated_hits_per_chunk(emu):
    """The emulation runs one lane at a time, so every hit is a group of its own (mask + entry): a chunk of
    kLogChunk slots, header included, holds (kLogChunk - 1) // 2 of them."""
    return (int(emu.lib.emu_log_chunk()) - 1) // 2


@pytest.mark.parametrize("sub_cap", [1024, 0, 1])
def test_emulated_radius_capture(sub_cap):
    """The capturing count pass + copy (RadiusCapture): rows equal the two-pass result whether the
    chunk pool is ample, absent (static chunk only) or runs dry (logs break off midway)."""
    pts, q = ds.uniform_cloud(20_000, 3, 41), ds.uniform_cloud(3_000, 3, 42)
    q[:200] = pts[:200]  # dense spots: a few long rows
    q = q[:2_990]        # the last wavefront is partly empty
    emu = EmulatedTree(pts, 10)
    ref = oracle.Oracle(pts, 10, "port")
    perm, _ = emu.morton_permutation(q)
    some_redone = False
    for radius, e in ((0.0004, None), (0.01, None), (0.03, None), (0.07, None), (0.02, 1.6)):
        want_off, want = ref.search_radius(q, radius, e=e)
        for pm in (None, perm):
            static_only = _rows_of_overfull_logs(want_off, pm, _emulated_hits_per_chunk(emu))
            off, got, redone = emu.search_radius_captured(q, radius, e=e, perm=pm, sub_cap=sub_cap)
            assert np.array_equal(off, want_off) and got.tobytes() == want.tobytes()
            if sub_cap == 1024:
                assert redone == 0
            elif sub_cap == 0:
                assert redone == static_only
            else:
                assert redone <= static_only
            some_redone |= redone > 0
    assert some_redone == (sub_cap != 1024)
    off, got, _ = emu.search_radius_captured(q, 0.03, sort=True, sub_cap=1024)
    _, want = ref.search_radius(q, 0.03, sort=True)
    assert np.array_equal(got["distance"], want["distance"])


@pytest.mark.parametrize("sub_cap", [1024, 0])
def test_emulated_radius_leaf_lists(sub_cap):
    """The radius search with the rows made from leaf lists (ptk_kernels_lists.hpp): the listing count pass, the
    replay through the ring (64 fibers per wavefront) -- rows equal the oracle's whether every list fits its chunks or
    the long ones are lost (static chunk only: 16 listed leaves per query) and go to the ordinary fill kernel."""
    pts, q = ds.uniform_cloud(20_000, 3, 41), ds.uniform_cloud(1_500, 3, 42)
    q[:200] = pts[:200]
    q = q[:1_470]        # the last wavefront is partly empty
    emu = EmulatedTree(pts, 10)
    ref = oracle.Oracle(pts, 10, "port")
    perm, _ = emu.morton_permutation(q)
    some_lost = False
    for radius, e in ((0.0004, None), (0.01, None), (0.03, None), (0.09, None), (0.02, 1.6)):
        want_off, want = ref.search_radius(q, radius, e=e)
        for pm in (None, perm):
            off, got, lost = emu.search_radius_lists(q, radius, e=e, perm=pm, sub_cap=sub_cap)
            assert np.array_equal(off, want_off) and got.tobytes() == want.tobytes()
            if sub_cap == 1024:
                assert lost == 0
            some_lost |= lost > 0
    assert some_lost == (sub_cap == 0)
    lidar, lq = ds.lidar_cloud(30_000, 1), ds.lidar_cloud(1_000, 2, pose=(3.0, 1.5))
    emu, ref = EmulatedTree(lidar, 40), oracle.Oracle(lidar, 40, "port")  # leaves of more than 32 points: listed in pieces
    want_off, want = ref.search_radius(lq, 1.0)
    off, got, lost = emu.search_radius_lists(lq, 1.0, sub_cap=1024)
    assert lost == 0 and np.array_equal(off, want_off) and got.tobytes() == want.tobytes()
    two = ds.uniform_cloud(5_000, 2, 7)
    emu, ref = EmulatedTree(two, 6), oracle.Oracle(two, 6, "port")
    want_off, want = ref.search_radius(two[:1_000], 0.002)
    off, got, lost = emu.search_radius_lists(two[:1_000], 0.002, sub_cap=8)
    assert np.array_equal(off, want_off) and got.tobytes() == want.tobytes()


@pytest.mark.parametrize("sub_cap", [1024, 0, 1])
def test_emulated_radius_capture_any_dimension(sub_cap):
    pts, q = ds.uniform_cloud(8_000, 5, 61), ds.uniform_cloud(1_000, 5, 62)
    emu = EmulatedTree(pts, 8)
    ref = oracle.Oracle(pts, 8, "port")
    perm = np.random.default_rng(5).permutation(len(q)).astype(np.uint32)  # any launch order, same rows
    for radius, e, pm in ((0.03, None, None), (0.12, None, perm), (0.1, 1.5, perm)):
        want_off, want = ref.search_radius(q, radius, e=e)
        off, got, redone = emu.search_radius_captured(q, radius, e=e, perm=pm, sub_cap=sub_cap)
        assert np.array_equal(off, want_off) and got.tobytes() == want.tobytes()
        static_only = _rows_of_overfull_logs(want_off, pm, _emulated_hits_per_chunk(emu))
        assert redone == (0 if sub_cap == 1024 else static_only if sub_cap == 0 else redone) and redone <= static_only
    assert want_off[-1] > 31 * len(q)


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference not present")
@pytest.mark.parametrize("metric", ["SO2", "SE2Squared"])
def test_emulated_topological_kernels_equal_the_compiled_reference(metric):
    """metric_so2 (points on the circle, 1-D) and metric_se2_squared (x, y, angle; 3-D) through
    traverse_topo of ptk_kernels_topo.hpp, against kd_tree<space, metric_so2 | metric_se2_squared>
    of the reference's own headers (oracle/_ref): wrap-around neighbours included."""
    import pico_tree_amd as pt
    if metric == "SO2":
        pts, q, leaf = ds.uniform_cloud(6_000, 1, 51), ds.uniform_cloud(1_500, 1, 52), 6
        q[:20] = np.float32(0.0005) * np.arange(20, dtype=np.float32)[:, None]          # next to the seam 0 ~ 1
        q[20:40] = np.float32(1.0) - np.float32(0.0005) * np.arange(20, dtype=np.float32)[:, None]
        radius = 0.002
    else:
        pts, q, leaf = ds.uniform_cloud(20_000, 3, 53), ds.uniform_cloud(1_500, 3, 54), 10
        q[:40, 2] = np.float32(0.001)
        q[40:80, 2] = np.float32(0.999)
        radius = 0.002
    emu = EmulatedTree(pts, leaf, pt.Metric[metric])
    ref = oracle.Oracle(pts, leaf, "reference", metric)
    perm = emu.morton_permutation(q)[0]
    for k, small in ((1, False), (1, True), (7, False), (7, True), (40, False)):
        want = ref.search_knn(q, k)
        for p in (None, perm):
            assert emu.search_knn(q, k, perm=p, small_stack=small).tobytes() == want.tobytes(), (k, small)
    assert emu.search_knn(q, 5, e=1.4).tobytes() == ref.search_knn(q, 5, e=1.4).tobytes()
    for kw in ({}, {"e": 1.5}):
        a, b = emu.search_radius(q, radius, **kw), ref.search_radius(q, radius, **kw)
        assert b[0][-1] > 0 and np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes()
    knn = ref.search_knn(q, 4)
    assert (np.abs(pts[knn["index"], -1] - q[:, -1:]) > 0.5).any()  # some neighbours are nearer through 0 ~ 1
