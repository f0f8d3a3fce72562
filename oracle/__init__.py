"""Test oracle for the batched k-NN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Two CPU implementations sit behind one small ctypes facade:

* ``Oracle("port")``      -> ``oracle/liboracle.so``: this repository's CPU
  restatement of the reference algorithm (``oracle/ptk_oracle.cpp``).
* ``Oracle("reference")`` -> ``oracle/_ref/libptk_ref.so``: the actual reference
  headers compiled in place from ``/root/reference`` (``oracle/ref_driver.cpp``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``pico_tree_amd``, ``include/``) never
does.  Nothing here reads ``/root/reference`` at run time -- the reference build
is a prebuilt shared object that travels with the repository snapshot.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_double, c_float, c_int, c_size_t, c_uint32, c_uint64, c_void_p, POINTER

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PORT_LIB = os.path.join(_HERE, "liboracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libptk_ref.so")
REF_FOREST_LIB = os.path.join(_HERE, "_ref", "libptk_ref_forest.so")
#: The same two libraries over double points (``-DPTKOR_DOUBLE`` / ``-DPTKREF_DOUBLE``).
PORT_LIB64 = os.path.join(_HERE, "liboracle64.so")
REF_LIB64 = os.path.join(_HERE, "_ref", "libptk_ref64.so")

#: Structured dtype of one result record; identical to the reference binding's
#: ``[('index','<i4'),('distance','<f4')]`` (_pyco_tree/def_core.hpp:17-18).
NEIGHBOR = np.dtype([("index", "<i4"), ("distance", "<f4")])
#: ``neighbor<int, double>``: 16 bytes, the distance at offset 8 (what pybind11's
#: PYBIND11_NUMPY_DTYPE yields for that struct, _pyco_tree/def_core.hpp:17-18).
NEIGHBOR64 = np.dtype({"names": ["index", "distance"], "formats": ["<i4", "<f8"], "offsets": [0, 8],
                       "itemsize": 16})


def build(force: bool = False) -> None:
    """Compile the oracle (and the reference driver when the sources exist)."""
    # make decides what is stale; the reference targets only exist where /root/reference does.
    if os.path.exists(os.path.join(_HERE, "Makefile")) and (force or _sources_present()):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))


def _sources_present() -> bool:
    return os.path.exists(os.path.join(_HERE, "ptk_oracle.cpp"))


def have_reference() -> bool:
    return os.path.exists(REF_LIB)


def have_reference64() -> bool:
    return os.path.exists(REF_LIB64)


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(POINTER(c_float))


class Oracle:
    """A CPU kd-tree over ``points`` answering batched queries.

    ``kind`` is ``"port"`` (the restatement) or ``"reference"`` (the compiled
    reference).  Results use the ``NEIGHBOR`` dtype; radius/box results are
    ``(offsets[nq+1], flat)`` pairs.
    """

    #: SO2 / SE2Squared: the reference's topological metrics, compiled reference only (the
    #: product runs them on the host members; tests/test_cpp_api.py).
    METRICS = {"L2Squared": 0, "L1": 1, "LPInf": 2, "SO2": 3, "SE2Squared": 4, "LNInf": 5}

    def __init__(self, points: np.ndarray, max_leaf_size: int = 10, kind: str = "port",
                 metric: str = "L2Squared", dtype=np.float32):
        if kind not in ("port", "reference"):
            raise ValueError(kind)
        self.kind = kind
        self.metric = metric
        mid = self.METRICS[metric]
        self.dtype = np.dtype(dtype)
        if self.dtype == np.float64:
            path = PORT_LIB64 if kind == "port" else REF_LIB64
            self._cs, self.neighbor = c_double, NEIGHBOR64
        elif self.dtype == np.float32:
            path = PORT_LIB if kind == "port" else REF_LIB
            self._cs, self.neighbor = c_float, NEIGHBOR
        else:
            raise ValueError("dtype must be float32 or float64")
        if not os.path.exists(path):
            raise RuntimeError(f"oracle library missing: {path} (run oracle.build())")
        self._lib = ctypes.CDLL(path)
        self._p = "ptkor_" if kind == "port" else "ptkref_"
        pts = np.ascontiguousarray(points, dtype=self.dtype)
        if pts.ndim != 2:
            raise ValueError("points must be (n, dim)")
        self.n, self.dim = pts.shape
        self.max_leaf_size = int(max_leaf_size)
        self._pts = pts
        if kind == "reference" and mid != 0:  # another instantiation of the reference's kd_tree
            create = self._fn("create_metric", c_void_p, [POINTER(self._cs), c_size_t, c_size_t, c_size_t, c_int])
            self._h = create(self._ptr(pts), self.n, self.dim, self.max_leaf_size, mid)
        else:
            create = self._fn("create", c_void_p, [POINTER(self._cs), c_size_t, c_size_t, c_size_t])
            self._h = create(self._ptr(pts), self.n, self.dim, self.max_leaf_size)
        if not self._h:
            raise RuntimeError("oracle create failed")
        if kind == "port" and mid in (3, 4):
            raise ValueError("the restatement covers the euclidean metrics only")
        if kind == "port" and mid != 0:
            if self._fn("set_metric", c_int, [c_void_p, c_int])(self._h, mid) != 0:
                raise RuntimeError("oracle set_metric failed")

    def _ptr(self, a: np.ndarray):
        return a.ctypes.data_as(POINTER(self._cs))

    def _fn(self, name, restype, argtypes):
        f = getattr(self._lib, self._p + name)
        f.restype = restype
        f.argtypes = argtypes
        return f

    def close(self):
        if getattr(self, "_h", None):
            self._fn("destroy", None, [c_void_p])(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- threads ------------------------------------------------------------
    def set_threads(self, n: int) -> None:
        self._fn("set_threads", None, [c_int])(int(n))

    def max_threads(self) -> int:
        return int(self._fn("max_threads", c_int, [])())

    # -- structure ----------------------------------------------------------
    def save_bytes(self) -> bytes:
        """The byte stream ``kd_tree::save`` writes for this tree."""
        f = self._fn("save", c_size_t, [c_void_p, c_void_p, c_size_t])
        size = f(self._h, None, 0)
        buf = np.empty(size, dtype=np.uint8)
        f(self._h, buf.ctypes.data, size)
        return buf.tobytes()

    def flatten(self):
        """(nodes uint32[n_nodes,4], indices int32[n], root_min, root_max, max_depth); port only."""
        if self.kind != "port":
            raise RuntimeError("flatten is a port-only helper")
        f = self._fn("flatten", c_size_t,
                     [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, POINTER(c_uint32)])
        count = f(self._h, None, 0, None, None, None, None)
        nodes = np.empty((count, 4), dtype=np.uint32)
        indices = np.empty(self.n, dtype=np.int32)
        rmin = np.empty(self.dim, dtype=self.dtype)
        rmax = np.empty(self.dim, dtype=self.dtype)
        depth = c_uint32(0)
        f(self._h, nodes.ctypes.data, count, indices.ctypes.data, rmin.ctypes.data,
          rmax.ctypes.data, ctypes.byref(depth))
        return nodes, indices, rmin, rmax, int(depth.value)

    # -- queries ------------------------------------------------------------
    def _queries(self, q):
        q = np.ascontiguousarray(q, dtype=self.dtype)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise ValueError("queries must be (nq, dim)")
        return q

    def search_nn(self, q, e: float | None = None, counters: bool = False):
        q = self._queries(q)
        out = np.empty(len(q), dtype=self.neighbor)
        if self.kind == "port":
            cnt = np.zeros((len(q), 5), dtype=np.uint32) if counters else None
            self._fn("search_nn", None,
                     [c_void_p, POINTER(self._cs), c_size_t, c_int, self._cs, c_void_p, c_void_p])(
                self._h, self._ptr(q), len(q), int(e is not None), float(e or 1.0),
                out.ctypes.data, cnt.ctypes.data if counters else None)
            return (out, cnt) if counters else out
        if e is not None or counters:
            raise RuntimeError("reference driver: use search_knn(k=1, e) / count_visits")
        self._fn("search_nn", None, [c_void_p, POINTER(self._cs), c_size_t, c_void_p])(
            self._h, self._ptr(q), len(q), out.ctypes.data)
        return out

    def search_knn(self, q, k: int, e: float | None = None, counters: bool = False):
        q = self._queries(q)
        k = int(k)
        if k < 1 or k > self.n:
            raise ValueError("oracle requires 1 <= k <= n")
        out = np.empty((len(q), k), dtype=self.neighbor)
        if self.kind == "port":
            cnt = np.zeros((len(q), 5), dtype=np.uint32) if counters else None
            self._fn("search_knn", None,
                     [c_void_p, POINTER(self._cs), c_size_t, c_size_t, c_int, self._cs,
                      c_void_p, c_void_p])(
                self._h, self._ptr(q), len(q), k, int(e is not None), float(e or 1.0),
                out.ctypes.data, cnt.ctypes.data if counters else None)
            return (out, cnt) if counters else out
        if counters:
            raise RuntimeError("reference driver: use count_visits")
        if e is None:
            self._fn("search_knn", None,
                     [c_void_p, POINTER(self._cs), c_size_t, c_size_t, c_void_p])(
                self._h, self._ptr(q), len(q), k, out.ctypes.data)
        else:
            self._fn("search_knn_approx", None,
                     [c_void_p, POINTER(self._cs), c_size_t, c_size_t, self._cs, c_void_p])(
                self._h, self._ptr(q), len(q), k, float(e), out.ctypes.data)
        return out

    def search_radius(self, q, radius: float, sort: bool = False, e: float | None = None,
                      counters: bool = False):
        q = self._queries(q)
        offsets = np.zeros(len(q) + 1, dtype=np.uint64)
        if self.kind == "port":
            cnt = np.zeros((len(q), 5), dtype=np.uint32) if counters else None
            h = self._fn("search_radius", c_void_p,
                         [c_void_p, POINTER(self._cs), c_size_t, self._cs, c_int, c_int, self._cs,
                          c_void_p, c_void_p])(
                self._h, self._ptr(q), len(q), float(radius), int(sort), int(e is not None),
                float(e or 1.0), offsets.ctypes.data, cnt.ctypes.data if counters else None)
        else:
            if counters:
                raise RuntimeError("reference driver has no radius counters")
            cnt = None
            h = self._fn("search_radius", c_void_p,
                         [c_void_p, POINTER(self._cs), c_size_t, self._cs, c_int, c_int, self._cs,
                          c_void_p])(
                self._h, self._ptr(q), len(q), float(radius), int(sort), int(e is not None),
                float(e or 1.0), offsets.ctypes.data)
        flat = np.empty(int(offsets[-1]), dtype=self.neighbor)
        self._fn("radius_copy", None, [c_void_p, c_void_p])(h, flat.ctypes.data)
        self._fn("radius_free", None, [c_void_p])(h)
        return (offsets, flat, cnt) if counters else (offsets, flat)

    def search_box(self, mins, maxs):
        mins = self._queries(mins)
        maxs = self._queries(maxs)
        offsets = np.zeros(len(mins) + 1, dtype=np.uint64)
        h = self._fn("search_box", c_void_p,
                     [c_void_p, POINTER(self._cs), POINTER(self._cs), c_size_t, c_void_p])(
            self._h, self._ptr(mins), self._ptr(maxs), len(mins), offsets.ctypes.data)
        flat = np.empty(int(offsets[-1]), dtype=np.int32)
        self._fn("box_copy", None, [c_void_p, c_void_p])(h, flat.ctypes.data)
        self._fn("box_free", None, [c_void_p])(h)
        return offsets, flat


def reference_count_visits(points, max_leaf_size, q, k):
    """Per-query (n_branch, n_pts) of the *reference* traversal (counting metric)."""
    lib = ctypes.CDLL(REF_LIB)
    pts = np.ascontiguousarray(points, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    nb = np.zeros(len(q), dtype=np.uint32)
    npts = np.zeros(len(q), dtype=np.uint32)
    f = lib.ptkref_count_visits
    f.restype = None
    f.argtypes = [POINTER(c_float), c_size_t, c_size_t, c_size_t, POINTER(c_float), c_size_t,
                  c_size_t, c_void_p, c_void_p]
    f(_fptr(pts), pts.shape[0], pts.shape[1], int(max_leaf_size), _fptr(q), len(q), int(k),
      nb.ctypes.data, npts.ctypes.data)
    return nb, npts


def sliding_midpoint_2d(kind, pts, indices, box_min, box_max):
    """Run the sliding-midpoint splitter KAT hook; returns (offset, dim, val, indices)."""
    lib = ctypes.CDLL(PORT_LIB if kind == "port" else REF_LIB)
    f = getattr(lib, ("ptkor_" if kind == "port" else "ptkref_") + "sliding_midpoint_2d")
    f.restype = None
    f.argtypes = [POINTER(c_float), c_size_t, c_void_p, POINTER(c_float), POINTER(c_float),
                  POINTER(c_size_t), POINTER(c_size_t), POINTER(c_float)]
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    idx = np.ascontiguousarray(indices, dtype=np.int32).copy()
    bmin = np.ascontiguousarray(box_min, dtype=np.float32)
    bmax = np.ascontiguousarray(box_max, dtype=np.float32)
    off, dim, val = c_size_t(0), c_size_t(0), c_float(0)
    f(_fptr(pts), len(pts), idx.ctypes.data, _fptr(bmin), _fptr(bmax),
      ctypes.byref(off), ctypes.byref(dim), ctypes.byref(val))
    return int(off.value), int(dim.value), float(val.value), idx


def l2sq(a, b) -> float:
    lib = ctypes.CDLL(PORT_LIB)
    lib.ptkor_l2sq.restype = c_float
    lib.ptkor_l2sq.argtypes = [POINTER(c_float), POINTER(c_float), c_size_t]
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib.ptkor_l2sq(_fptr(a), _fptr(b), len(a)))


def distance(metric: str, a, b) -> float:
    lib = ctypes.CDLL(PORT_LIB)
    lib.ptkor_distance.restype = c_float
    lib.ptkor_distance.argtypes = [c_int, POINTER(c_float), POINTER(c_float), c_size_t]
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib.ptkor_distance(Oracle.METRICS[metric], _fptr(a), _fptr(b), len(a)))


def distance_scalar(metric: str, x: float) -> float:
    lib = ctypes.CDLL(PORT_LIB)
    lib.ptkor_distance_scalar.restype = c_float
    lib.ptkor_distance_scalar.argtypes = [c_int, c_float]
    return float(lib.ptkor_distance_scalar(Oracle.METRICS[metric], x))


def l2sq_scalar(x: float) -> float:
    lib = ctypes.CDLL(PORT_LIB)
    lib.ptkor_l2sq_scalar.restype = c_float
    lib.ptkor_l2sq_scalar.argtypes = [c_float]
    return float(lib.ptkor_l2sq_scalar(x))


# ---- kd_forest ------------------------------------------------------------------------------

class ForestOracle:
    """CPU restatement of the forest search the product implements (oracle/ptk_oracle.cpp,
    section kd_forest): same reflections (passed in), same de-duplicated k-list, distances in the
    original space.  The GPU result must equal this bit for bit."""

    def __init__(self, points: np.ndarray, max_leaf_size: int, rotations: np.ndarray):
        build()
        self._lib = ctypes.CDLL(PORT_LIB)
        self._pts = np.ascontiguousarray(points, dtype=np.float32)
        rot = np.ascontiguousarray(rotations, dtype=np.float32)
        n, dim = self._pts.shape
        assert rot.ndim == 2 and rot.shape[1] == dim
        self._dim = dim
        fn = self._lib.ptkor_forest_create
        fn.restype = c_void_p
        fn.argtypes = [c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]
        self._h = fn(self._pts.ctypes.data, n, dim, int(max_leaf_size), rot.shape[0], rot.ctypes.data)
        assert self._h

    def search_knn(self, q: np.ndarray, k: int, max_leaves_visited: int) -> np.ndarray:
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.empty((len(q), k), dtype=NEIGHBOR)
        fn = self._lib.ptkor_forest_search_knn
        fn.restype = None
        fn.argtypes = [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]
        fn(self._h, q.ctypes.data, len(q), int(k), int(max_leaves_visited), out.ctypes.data)
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ptkor_forest_destroy.argtypes = [c_void_p]
            self._lib.ptkor_forest_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def have_reference_forest() -> bool:
    return os.path.exists(REF_FOREST_LIB)


class ReferenceForest:
    """The compiled reference kd_forest (oracle/ref_forest_driver.cpp).  Its reflections come
    from std::random_device, so only statistics (recall, speed) are comparable."""

    def __init__(self, points: np.ndarray, max_leaf_size: int, forest_size: int):
        self._lib = ctypes.CDLL(REF_FOREST_LIB)
        self._pts = np.ascontiguousarray(points, dtype=np.float32)
        n, dim = self._pts.shape
        self._dim = dim
        fn = self._lib.ptkref_forest_create
        fn.restype = c_void_p
        fn.argtypes = [c_void_p, c_size_t, c_size_t, c_size_t, c_size_t]
        self._h = fn(self._pts.ctypes.data, n, dim, int(max_leaf_size), int(forest_size))
        assert self._h

    def search_knn(self, q: np.ndarray, k: int, max_leaves_visited: int) -> np.ndarray:
        q = np.ascontiguousarray(q, dtype=np.float32)
        out = np.zeros((len(q), k), dtype=NEIGHBOR)
        fn = self._lib.ptkref_forest_search_knn
        fn.restype = None
        fn.argtypes = [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_size_t, c_void_p]
        fn(self._h, q.ctypes.data, len(q), self._dim, int(k), int(max_leaves_visited), out.ctypes.data)
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ptkref_forest_destroy.argtypes = [c_void_p]
            self._lib.ptkref_forest_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def canonical_stream64(stream: bytes) -> bytes:
    """A double-scalar ``kd_tree::save`` stream with the 4 padding bytes of every branch record
    zeroed.  The reference writes ``kd_tree_branch_single<double>`` whole
    (internal/kd_tree_data.hpp:116): {int split_dim; <4 bytes padding>; double; double}, and the
    padding holds whatever the allocator left there."""
    import struct

    b = bytearray(stream)
    sdim, n = struct.unpack_from("<QQ", b, 0)
    pos = 16 + 4 * n + 16 * sdim
    while pos < len(b):
        leaf = b[pos]
        pos += 1
        if leaf:
            pos += 8
        else:
            b[pos + 4:pos + 8] = b"\0\0\0\0"
            pos += 24
    if pos != len(b):
        raise ValueError("not a double-scalar kd_tree stream")
    return bytes(b)
