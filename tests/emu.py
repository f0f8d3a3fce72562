"""ctypes facade over tests/cpp/libptk_emu.so: the product's kernel source
(pico_tree_amd/csrc/ptk_kernels.hpp) compiled for the host and run lane by lane."""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_uint32, c_uint64, c_void_p

import numpy as np

import pico_tree_amd as pt

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "libptk_emu.so")


def _lib():
    lib = ctypes.CDLL(_LIB)
    lib.emu_create.restype = c_void_p
    lib.emu_create.argtypes = [c_void_p, c_uint64, c_uint32, c_void_p, c_uint64, c_void_p]
    lib.emu_destroy.argtypes = [c_void_p]
    lib.emu_last_error.restype = c_char_p
    lib.emu_max_depth.restype = c_uint32
    lib.emu_max_depth.argtypes = [c_void_p]
    lib.emu_knn.argtypes = [c_void_p, c_void_p, c_uint64, c_uint32, c_float, c_void_p, c_int, c_int, c_void_p]
    lib.emu_radius_count.argtypes = [c_void_p, c_void_p, c_uint64, c_float, c_float, c_void_p, c_void_p]
    lib.emu_radius_fill.argtypes = [c_void_p, c_void_p, c_uint64, c_float, c_float, c_void_p, c_void_p,
                                    c_void_p, c_int]
    lib.emu_knn1_two_phase.argtypes = [c_void_p, c_void_p, c_uint64, c_float, c_void_p, c_int, c_void_p]
    lib.emu_morton.argtypes = [c_void_p, c_uint32, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    return lib


class EmulatedTree:
    """Builds the flat tree with the PRODUCT builder (host-only libptk handle),
    encodes it with the product encoder and runs the product kernels on the CPU."""

    def __init__(self, pts, leaf, metric=pt.Metric.L2Squared):
        self.lib = _lib()
        self.pts = np.ascontiguousarray(pts, dtype=np.float32)
        self.leaf = int(leaf)
        # (a host-only handle: the topological metrics are set on the emulator, the tree is the same)
        self.host = pt.KdTree(self.pts, pt.Metric.L2Squared if metric.name in ("SO2", "SE2Squared") else metric, leaf,
                              device=pt.PTK_DEVICE_NONE)
        nodes, idx, self.rmin, self.rmax = self.host.flat()
        self.h = self.lib.emu_create(self.pts.ctypes.data, len(self.pts), self.pts.shape[1],
                                     nodes.ctypes.data, len(nodes), idx.ctypes.data)
        if not self.h:
            raise RuntimeError(self.lib.emu_last_error().decode())
        self.lib.emu_set_metric.argtypes = [c_void_p, ctypes.c_int]
        self.lib.emu_set_metric.restype = None
        mid = {"L2Squared": 0, "L1": 1, "LPInf": 2, "LNInf": 3, "SO2": 4, "SE2Squared": 5}[metric.name]
        if mid >= 4:  # the two outer bounds per branch, as the host builder left them
            outer = np.empty((len(nodes), 2), dtype=np.float32)
            assert pt._load().ptk_tree_get_outer_bounds(self.host._h, outer.ctypes.data) == 0
            self.lib.emu_set_outer.argtypes = [c_void_p, c_void_p, c_uint64, c_uint64, c_void_p]
            assert self.lib.emu_set_outer(self.h, nodes.ctypes.data, len(nodes), len(self.pts), outer.ctypes.data) == 0
        self.lib.emu_set_metric(self.h, mid)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.emu_destroy(self.h)
            self.h = None

    def search_knn(self, q, k, e=None, perm=None, small_stack=False, list_in_lds=True):
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.zeros((len(q), k), dtype=pt.NEIGHBOR)
        rc = self.lib.emu_knn(self.h, q.ctypes.data, len(q), k, e or 1.0,
                              perm.ctypes.data if perm is not None else None,
                              int(small_stack), int(list_in_lds), out.ctypes.data)
        assert rc == 0
        return out

    def search_knn_capped(self, q, k, cap, perm=None, pool_small=False, max_heavy=None):
        """The general k-NN kernel capped at `cap` far children per query, the cooperative search of what it handed
        over and the reference search of what that could not certify (ptk_kernels_coopk.hpp).
        Returns (rows, queries handed over, queries redone); `last_tie_sweeps` = queries that took the second sweep."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.zeros((len(q), k), dtype=pt.NEIGHBOR)
        counts = np.zeros(3, dtype=np.uint32)
        fn = self.lib.emu_knn_capped
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_uint64, c_uint32, c_void_p, c_uint32, c_int, c_uint32, c_void_p, c_void_p]
        rc = fn(self.h, q.ctypes.data, len(q), k, perm.ctypes.data if perm is not None else None, cap,
                int(pool_small), len(q) if max_heavy is None else max_heavy, out.ctypes.data, counts.ctypes.data)
        assert rc == 0
        self.last_tie_sweeps = int(counts[2])
        return out, int(counts[0]), int(counts[1])

    def search_radius(self, q, radius, sort=False, e=None, perm=None):
        q = np.ascontiguousarray(q, dtype=np.float32)
        nq = len(q)
        p = perm.ctypes.data if perm is not None else None
        counts = np.zeros(nq + 1, dtype=np.uint64)
        self.lib.emu_radius_count(self.h, q.ctypes.data, nq, radius, e or 1.0, p, counts.ctypes.data)
        off = np.zeros(nq + 1, dtype=np.uint64)
        off[1:] = np.cumsum(counts[:nq])
        out = np.zeros(int(off[-1]), dtype=pt.NEIGHBOR)
        self.lib.emu_radius_fill(self.h, q.ctypes.data, nq, radius, e or 1.0, p, off.ctypes.data,
                                 out.ctypes.data, int(sort))
        return off, out

    def search_radius_captured(self, q, radius, sort=False, e=None, perm=None, sub_cap=64):
        """Count pass with capture, then the copy (+ re-traversal of rows that did not fit)."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        nq = len(q)
        p = perm.ctypes.data if perm is not None else None
        counts = np.zeros(nq + 1, dtype=np.uint64)
        self.lib.emu_radius_capture.argtypes = [c_void_p, c_void_p, c_uint64, c_float, c_float, c_void_p, c_void_p,
                                                c_uint32]
        assert self.lib.emu_radius_capture(self.h, q.ctypes.data, nq, radius, e or 1.0, p, counts.ctypes.data,
                                           sub_cap) == 0
        off = np.zeros(nq + 1, dtype=np.uint64)
        off[1:] = np.cumsum(counts[:nq])
        out = np.zeros(int(off[-1]), dtype=pt.NEIGHBOR)
        self.lib.emu_radius_fill_captured.restype = ctypes.c_int64
        self.lib.emu_radius_fill_captured.argtypes = [c_void_p, c_void_p, c_uint64, c_float, c_float, c_void_p,
                                                      c_void_p, c_int]
        redone = self.lib.emu_radius_fill_captured(self.h, q.ctypes.data, nq, radius, e or 1.0, off.ctypes.data,
                                                   out.ctypes.data, int(sort))
        assert redone >= 0
        return off, out, int(redone)

    def search_radius_lists(self, q, radius, sort=False, e=None, perm=None, sub_cap=64):
        """The radius search with the rows made from leaf lists: the listing count pass, then the replay
        (+ the ordinary fill kernel for wavefronts whose lists were lost)."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        nq = len(q)
        p = perm.ctypes.data if perm is not None else None
        counts = np.zeros(nq + 1, dtype=np.uint64)
        fn = self.lib.emu_radius_lists
        fn.restype = ctypes.c_int64
        fn.argtypes = [c_void_p, c_void_p, c_uint64, c_float, c_float, c_void_p, c_uint32, c_int, c_void_p, c_void_p,
                       c_void_p]
        assert fn(self.h, q.ctypes.data, nq, radius, e or 1.0, p, sub_cap, 0, counts.ctypes.data, None, None) == 0
        off = np.zeros(nq + 1, dtype=np.uint64)
        off[1:] = np.cumsum(counts[:nq])
        out = np.zeros(int(off[-1]), dtype=pt.NEIGHBOR)
        lost = fn(self.h, q.ctypes.data, nq, radius, e or 1.0, p, sub_cap, 1, None, off.ctypes.data, out.ctypes.data)
        assert lost >= 0
        return off, out, int(lost)

    def search_radius_lists_capped(self, q, radius, far_cap, e=None, perm=None, sub_cap=64, pool_small=False,
                                   max_heavy=None, entry_cap=None):
        """The radius search with its long queries finished by a wavefront each (ptk_kernels_coopr.hpp): the list pass
        capped at `far_cap` far children per query, the cooperative count, the recount of what was lost; the replay, the
        cooperative replay, the ordinary fill kernel for what was lost.
        Returns (offsets, rows, {handed over, recounted from the root, rows filled by the ordinary kernel})."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        nq = len(q)
        p = perm.ctypes.data if perm is not None else None
        counts = np.zeros(nq + 1, dtype=np.uint64)
        mh = nq if max_heavy is None else max_heavy
        ec = mh * 1024 if entry_cap is None else entry_cap
        fn = self.lib.emu_radius_lists_capped
        fn.restype = ctypes.c_int64
        fn.argtypes = [c_void_p, c_void_p, c_uint64, c_float, c_float, c_void_p, c_uint32, c_uint32, c_int, c_uint32,
                       c_uint32, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
        args = (self.h, q.ctypes.data, nq, radius, e or 1.0, p, sub_cap, far_cap, int(pool_small), mh, ec)
        assert fn(*args, 0, counts.ctypes.data, None, None, None) == 0
        off = np.zeros(nq + 1, dtype=np.uint64)
        off[1:] = np.cumsum(counts[:nq])
        out = np.zeros(int(off[-1]), dtype=pt.NEIGHBOR)
        stats = np.zeros(3, dtype=np.uint32)
        assert fn(*args, 1, None, off.ctypes.data, out.ctypes.data, stats.ctypes.data) >= 0
        return off, out, {"handed_over": int(stats[0]), "recounted": int(stats[1]), "refilled": int(stats[2])}

    # -- persistent (state machine + lane refill) kernels: 64 host threads per wavefront --
    def search_box(self, mins, maxs):
        from ctypes import c_uint64, c_void_p
        mins = np.ascontiguousarray(mins, dtype=np.float32)
        maxs = np.ascontiguousarray(maxs, dtype=np.float32)
        nb = len(mins)
        rmin = np.ascontiguousarray(self.pts.min(0), dtype=np.float32)
        rmax = np.ascontiguousarray(self.pts.max(0), dtype=np.float32)
        off = np.zeros(nb + 1, dtype=np.uint64)
        self.lib.emu_box.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]
        assert self.lib.emu_box(self.h, mins.ctypes.data, maxs.ctypes.data, nb, rmin.ctypes.data, rmax.ctypes.data,
                                off.ctypes.data, None) == 0
        out = np.zeros(max(int(off[-1]), 1), dtype=np.int32)
        assert self.lib.emu_box(self.h, mins.ctypes.data, maxs.ctypes.data, nb, rmin.ctypes.data, rmax.ctypes.data,
                                off.ctypes.data, out.ctypes.data) == 0
        return off, out[:int(off[-1])]

    def use_pile_view(self):
        """Switches the handle to the k = 1 view of its tree (ptk_piles.hpp); returns the number of piles."""
        nodes, idx, _, _ = self.host.flat()
        fn = self.lib.emu_use_pile_view
        fn.restype = ctypes.c_int64
        fn.argtypes = [c_void_p, c_void_p, c_uint64, c_void_p, c_uint64, c_void_p]
        n = fn(self.h, self.pts.ctypes.data, len(self.pts), nodes.ctypes.data, len(nodes), idx.ctypes.data)
        assert n >= 0
        self._piles = int(n)
        return self._piles

    def two_phase_knn1(self, q, e=None, perm=None, variant=5):
        """Returns (result (nq,1), number of continuations handed to phase 2)."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.zeros((len(q), 1), dtype=pt.NEIGHBOR)
        n2 = self.lib.emu_knn1_two_phase(self.h, q.ctypes.data, len(q), e or 1.0,
                                         perm.ctypes.data if perm is not None else None, variant,
                                         out.ctypes.data)
        assert n2 >= 0
        if getattr(self, "_piles", 0):
            self.lib.emu_resolve_piles.restype = None
            self.lib.emu_resolve_piles.argtypes = [c_void_p, c_void_p, c_uint64, c_void_p]
            self.lib.emu_resolve_piles(self.h, q.ctypes.data, len(q), out.ctypes.data)
        return out, n2

    def last_coop(self):
        """(queries phase 2 gave up on, queries the cooperative search could not certify) of the last
        ``two_phase_knn1`` call with a capped variant (5-8)."""
        heavy, redo = c_uint32(0), c_uint32(0)
        self.lib.emu_last_coop(ctypes.byref(heavy), ctypes.byref(redo))
        return heavy.value, redo.value

    def last_spilled(self):
        """Spill slots (HBM) the cooperative launches of the last variant-9 ``two_phase_knn1`` call wrote."""
        self.lib.emu_last_spilled.restype = c_uint32
        return int(self.lib.emu_last_spilled())

    def morton_permutation(self, q, bits=None):
        """A permutation as the device makes it (keys from the kernel, stable sort).  `bits`: key bits per
        axis (the backend derives them from the tree, ptk_backend.hip axis_bits(); default 8 each)."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        bits = np.zeros(3, dtype=np.uint32) if bits is None else np.asarray(list(bits) + [0] * 3, dtype=np.uint32)[:3]
        if not bits.any():
            bits[:self.pts.shape[1]] = 8
        lo = np.zeros(3, dtype=np.float32)
        inv = np.zeros(3, dtype=np.float32)
        d = self.pts.shape[1]
        lo[:d] = self.rmin
        ext = self.rmax - self.rmin
        inv[:d] = np.where(ext > 0, np.exp2(bits[:d]).astype(np.float32) / ext, 0).astype(np.float32)
        keys = np.zeros(len(q), dtype=np.uint32)
        ids = np.zeros(len(q), dtype=np.uint32)
        self.lib.emu_morton(q.ctypes.data, d, len(q), lo.ctypes.data, inv.ctypes.data, bits.ctypes.data,
                            keys.ctypes.data, ids.ctypes.data)
        return ids[np.argsort(keys, kind="stable")].astype(np.uint32), keys

    def radix_sorted_permutation(self, q, bits, tile=256):
        """The batch order from the library's own radix sort (ptk_sort.hpp) under the emulator:
        (permutation, Morton keys).  `bits`: key bits per axis.  tile = 0: the passes with blocks of four wavefronts
        on tiles of 4 096 items."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        bits = np.asarray(list(bits) + [0] * 3, dtype=np.uint32)[:3]
        lo = np.zeros(3, dtype=np.float32)
        inv = np.zeros(3, dtype=np.float32)
        d = self.pts.shape[1]
        lo[:d] = self.rmin
        ext = self.rmax - self.rmin
        inv[:d] = np.where(ext > 0, np.exp2(bits[:d]).astype(np.float32) / ext, 0).astype(np.float32)
        keys = np.zeros(len(q), dtype=np.uint32)
        perm = np.zeros(len(q), dtype=np.uint32)
        if tile == 0:
            self.lib.emu_radix_sort_blocks.argtypes = [c_void_p, c_uint32, c_uint64, c_void_p, c_void_p, c_void_p, c_uint32,
                                                       c_void_p, c_void_p]
            self.lib.emu_radix_sort_blocks.restype = None
            self.lib.emu_radix_sort_blocks(q.ctypes.data, d, len(q), lo.ctypes.data, inv.ctypes.data, bits.ctypes.data,
                                           int(bits.sum()), keys.ctypes.data, perm.ctypes.data)
            return perm, keys
        self.lib.emu_radix_sort.argtypes = [c_void_p, c_uint32, c_uint64, c_void_p, c_void_p, c_void_p, c_uint32,
                                            c_uint32, c_void_p, c_void_p]
        self.lib.emu_radix_sort.restype = None
        self.lib.emu_radix_sort(q.ctypes.data, d, len(q), lo.ctypes.data, inv.ctypes.data, bits.ctypes.data,
                                int(bits.sum()), int(tile), keys.ctypes.data, perm.ctypes.data)
        return perm, keys


def emulated_forest_knn(pts, max_leaf, n_trees, seed, q, k, max_leaves):
    """Host build (product code) + forest kernel under the CPU emulator.
    Returns (result (nq, k), rotations (n_trees, dim), dropped queue entries)."""
    from ctypes import c_uint32, c_uint64, c_void_p
    lib = _lib()
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    n, dim = pts.shape
    rot = np.zeros((n_trees, dim), dtype=np.float32)
    out = np.zeros((len(q), k), dtype=pt.NEIGHBOR)
    dropped = np.zeros(1, dtype=np.uint32)
    lib.emu_forest_knn.argtypes = [c_void_p, c_uint64, c_uint32, c_uint64, c_uint32, c_uint64, c_void_p, c_uint64,
                                   c_uint32, c_uint32, c_void_p, c_void_p, c_void_p]
    rc = lib.emu_forest_knn(pts.ctypes.data, n, dim, max_leaf, n_trees, seed, q.ctypes.data, len(q), k, max_leaves,
                            rot.ctypes.data, out.ctypes.data, dropped.ctypes.data)
    assert rc == 0, lib.emu_last_error()
    return out, rot, int(dropped[0])


class EmulatedTree64:
    """The double-precision kernels (pico_tree_amd/csrc/ptk_kernels_f64.hpp) run lane by lane on
    the CPU over a tree built by the product's host builder instantiated over double."""

    def __init__(self, pts, leaf, metric="L2Squared"):
        import oracle

        self.neighbor = oracle.NEIGHBOR64
        self.lib = ctypes.CDLL(_LIB)
        self.lib.emu64_create.restype = c_void_p
        self.lib.emu64_create.argtypes = [c_void_p, c_uint64, c_uint32, c_uint64]
        self.lib.emu64_destroy.argtypes = [c_void_p]
        self.lib.emu64_set_metric.argtypes = [c_void_p, c_int]
        self.lib.emu64_save.restype = c_uint64
        self.lib.emu64_save.argtypes = [c_void_p, c_void_p, c_uint64]
        self.lib.emu64_knn.argtypes = [c_void_p, c_void_p, c_uint64, c_uint32, ctypes.c_double, c_int, c_void_p]
        self.lib.emu64_radius.argtypes = [c_void_p, c_void_p, c_uint64, ctypes.c_double, ctypes.c_double, c_int,
                                          c_void_p, c_void_p]
        self.lib.emu64_box.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p]
        self.pts = np.ascontiguousarray(pts, dtype=np.float64)
        self.h = self.lib.emu64_create(self.pts.ctypes.data, len(self.pts), self.pts.shape[1], int(leaf))
        if not self.h:
            raise RuntimeError(self.lib.emu_last_error().decode())
        self.lib.emu64_set_metric(self.h, {"L2Squared": 0, "L1": 1, "LPInf": 2, "LNInf": 3, "SO2": 4, "SE2Squared": 5}[metric])

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.emu64_destroy(self.h)
            self.h = None

    def save_bytes(self) -> bytes:
        size = self.lib.emu64_save(self.h, None, 0)
        buf = np.empty(size, dtype=np.uint8)
        self.lib.emu64_save(self.h, buf.ctypes.data, size)
        return buf.tobytes()

    def search_knn(self, q, k, e=None, list_in_registers=True):
        q = np.ascontiguousarray(q, dtype=np.float64)
        out = np.zeros((len(q), k), dtype=self.neighbor)
        assert self.lib.emu64_knn(self.h, q.ctypes.data, len(q), k, e or 1.0, int(list_in_registers),
                                  out.ctypes.data) == 0
        return out

    def search_knn_capped(self, q, k, cap, perm=None, pool_small=False, max_heavy=None):
        """The capped double k-NN kernel, the cooperative search of what it handed over and the reference search of what
        that could not certify (ptk_kernels_coop64.hpp).  Returns (rows, queries handed over, queries redone);
        `last_tie_sweeps` = queries that took the second sweep."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        out = np.zeros((len(q), k), dtype=self.neighbor)
        counts = np.zeros(3, dtype=np.uint32)
        fn = self.lib.emu64_knn_capped
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_uint64, c_uint32, c_void_p, c_uint32, c_int, c_uint32, c_void_p, c_void_p]
        if perm is not None:
            perm = np.ascontiguousarray(perm, dtype=np.uint32)
        rc = fn(self.h, q.ctypes.data, len(q), k, perm.ctypes.data if perm is not None else None, cap,
                int(pool_small), len(q) if max_heavy is None else max_heavy, out.ctypes.data, counts.ctypes.data)
        assert rc == 0
        self.last_tie_sweeps = int(counts[2])
        return out, int(counts[0]), int(counts[1])

    def search_radius_capped(self, q, radius, cap, e=None, small=False, max_heavy=None):
        """Both passes of the capped double radius search (ptk_kernels_coop64.hpp): the capped count launch, the cooperative
        count of what it handed over, the recount of what that lost; the capped fill launch, the cooperative replay, the
        refill of the lost rows.  Returns (offsets, rows, queries handed over, rows searched again from the root)."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        off = np.zeros(len(q) + 1, dtype=np.uint64)
        counts = np.zeros(3, dtype=np.uint32)
        fn = self.lib.emu64_radius_capped
        fn.restype = c_int
        fn.argtypes = [c_void_p, c_void_p, c_uint64, ctypes.c_double, ctypes.c_double, c_uint32, c_int, c_uint32, c_void_p,
                       c_void_p, c_void_p]
        mh = len(q) if max_heavy is None else max_heavy
        assert fn(self.h, q.ctypes.data, len(q), radius, e or 1.0, cap, int(small), mh, off.ctypes.data, None,
                  counts.ctypes.data) == 0
        flat = np.zeros(int(off[-1]), dtype=self.neighbor)
        assert fn(self.h, q.ctypes.data, len(q), radius, e or 1.0, cap, int(small), mh, off.ctypes.data, flat.ctypes.data,
                  counts.ctypes.data) == 0
        return off, flat, int(counts[0]), int(counts[1])

    def search_radius(self, q, radius, e=None, sort=False):
        q = np.ascontiguousarray(q, dtype=np.float64)
        off = np.zeros(len(q) + 1, dtype=np.uint64)
        assert self.lib.emu64_radius(self.h, q.ctypes.data, len(q), radius, e or 1.0, 0, off.ctypes.data, None) == 0
        flat = np.zeros(int(off[-1]), dtype=self.neighbor)
        assert self.lib.emu64_radius(self.h, q.ctypes.data, len(q), radius, e or 1.0, int(sort), off.ctypes.data,
                                     flat.ctypes.data) == 0
        return off, flat

    def search_box(self, mins, maxs):
        mins = np.ascontiguousarray(mins, dtype=np.float64)
        maxs = np.ascontiguousarray(maxs, dtype=np.float64)
        off = np.zeros(len(mins) + 1, dtype=np.uint64)
        assert self.lib.emu64_box(self.h, mins.ctypes.data, maxs.ctypes.data, len(mins), off.ctypes.data, None) == 0
        flat = np.zeros(int(off[-1]), dtype=np.int32)
        assert self.lib.emu64_box(self.h, mins.ctypes.data, maxs.ctypes.data, len(mins), off.ctypes.data,
                                  flat.ctypes.data) == 0
        return off, flat
