"""pico_tree_amd -- Python host side of the MI355X batched k-NN backend.

Mirrors the reference's Python module (``pico_tree.KdTree``,
``/root/reference/src/pyco_tree/pico_tree/_pyco_tree/def_kd_tree.cpp:14-195``):
same constructor arguments, ``search_knn`` / ``search_radius`` overloads, result
dtype ``[('index','<i4'),('distance','<f4')]`` and output shapes
(``(npts,)`` for k == 1, ``(npts, k)`` otherwise;
``_pyco_tree/kd_tree.hpp:362-378``).  Every search runs on the GPU through the
C ABI of ``include/ptk.h`` (``pico_tree_amd/csrc/libptk.so``); there is no CPU
search path here and no fallback: if the library or a device is missing the
call raises.

Two kinds of buffers are accepted:

* numpy arrays (host): copied to the device and back by the backend;
* torch CUDA tensors (device): used in place on the current torch stream, the
  result is a :class:`DeviceNeighbors` holding an ``int32`` tensor of shape
  ``(nq, k, 2)`` whose last axis is ``(index, bits of the float32 distance)``.
"""

from __future__ import annotations

import ctypes
import enum
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int32,
                    c_uint32, c_uint64, c_void_p)

import numpy as np

from . import datasets  # noqa: F401  (re-export)

__all__ = ["KdTree", "KdForest", "save_kd_tree", "load_kd_tree", "Metric", "NEIGHBOR", "NEIGHBOR64", "DArray", "DeviceNeighbors", "PtkError",
           "library_path", "device_count", "datasets", "trim_pinned_pool", "registered", "empty_pinned"]

_HERE = os.path.dirname(os.path.abspath(__file__))
#: (PTK_LIBRARY: another build of the same library -- the experiment builds of tools/)
_LIB_PATH = os.environ.get("PTK_LIBRARY") or os.path.join(_HERE, "csrc", "libptk.so")

#: Result record; identical to the reference binding's neighbor dtype.
NEIGHBOR = np.dtype([("index", "<i4"), ("distance", "<f4")])
#: The same record of a tree over float64 points: ``neighbor<int, double>`` is 16 bytes with the
#: distance at offset 8, which is what the reference binding's PYBIND11_NUMPY_DTYPE yields for it
#: (_pyco_tree/def_core.hpp:17-18).
NEIGHBOR64 = np.dtype({"names": ["index", "distance"], "formats": ["<i4", "<f8"], "offsets": [0, 8],
                       "itemsize": 16})

PTK_OK = 0
PTK_DEVICE_CURRENT = -1
PTK_DEVICE_NONE = -2
REORDER_AUTO, REORDER_ON, REORDER_OFF = 0, 1, 2


class PtkError(RuntimeError):
    """A non-zero status from libptk (message from ``ptk_last_error``)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libptk status {status}: {message}")
        self.status = status


class _Info(Structure):
    _fields_ = [("dim", c_uint32), ("n_points", c_uint64), ("n_nodes", c_uint64),
                ("n_leaves", c_uint64), ("max_depth", c_uint32), ("max_leaf_count", c_uint32),
                ("device_bytes", c_uint64), ("device", c_int32)]


class _Profile(Structure):
    _fields_ = [("search_ms", c_double), ("reorder_ms", c_double), ("other_ms", c_double),
                ("launches", c_uint64), ("queries", c_uint64), ("search_tail_ms", c_double)]


_lib = None

_SIGNATURES = {
    "ptk_version": (c_int, []),
    "ptk_last_error": (c_char_p, []),
    "ptk_device_count": (c_int, []),
    "ptk_warmup": (c_int, [c_int32]),
    "ptk_tree_create_from_points": (c_int, [c_void_p, c_uint64, c_uint32, c_uint64, c_int32,
                                            POINTER(c_void_p)]),
    "ptk_tree_create": (c_int, [c_void_p, POINTER(c_void_p)]),
    "ptk_tree_destroy": (None, [c_void_p]),
    "ptk_tree_get_info": (c_int, [c_void_p, POINTER(_Info)]),
    "ptk_tree_get_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ptk_tree_set_reorder": (c_int, [c_void_p, c_int]),
    "ptk_tree_set_metric": (c_int, [c_void_p, c_int]),
    "ptk_tree_set_outer_bounds": (c_int, [c_void_p, c_void_p, c_uint64]),
    "ptk_tree_get_outer_bounds": (c_int, [c_void_p, c_void_p]),
    "ptk_tree_serialize": (c_int, [c_void_p, c_void_p, c_uint64, POINTER(c_uint64)]),
    "ptk_tree_create_from_stream": (c_int, [c_void_p, c_uint64, c_uint32, c_void_p, c_uint64, c_int32,
                                            POINTER(c_void_p)]),
    "ptk_tree_create_from_topological_stream": (c_int, [c_void_p, c_uint64, c_uint32, c_void_p, c_uint64, c_int32,
                                                        POINTER(c_void_p)]),
    "ptk_tree_serialize_topological": (c_int, [c_void_p, c_void_p, c_uint64, POINTER(c_uint64)]),
    "ptk_search_knn": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_float, c_void_p]),
    "ptk_search_knn_device": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_float, c_void_p,
                                      c_void_p]),
    "ptk_search_radius_count": (c_int, [c_void_p, c_void_p, c_uint64, c_float, c_float, c_void_p]),
    "ptk_search_radius_fill": (c_int, [c_void_p, c_void_p, c_uint64, c_float, c_float, c_void_p,
                                       c_void_p, c_int]),
    "ptk_search_radius_count_device": (c_int, [c_void_p, c_void_p, c_uint64, c_float, c_float,
                                               c_void_p, c_void_p]),
    "ptk_search_radius_fill_device": (c_int, [c_void_p, c_void_p, c_uint64, c_float, c_float,
                                              c_void_p, c_void_p, c_int, c_void_p]),
    "ptk_search_radius": (c_int, [c_void_p, c_void_p, c_uint64, c_float, c_float, c_int, c_void_p,
                                  POINTER(c_void_p)]),
    "ptk_search_box": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p,
                               POINTER(c_void_p)]),
    "ptk_search_box_count_device": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p]),
    "ptk_search_box_fill_device": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p]),
    "ptk_free": (None, [c_void_p]),
    "ptk_host_search_knn": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_uint32, c_float, c_void_p]),
    "ptk_host_search_radius": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_float, c_float, c_int,
                                       c_void_p, POINTER(c_void_p)]),
    "ptk_host_search_box": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, POINTER(c_void_p)]),
    "ptk_host_alloc": (c_int, [c_uint64, POINTER(c_void_p)]),
    "ptk_host_free": (None, [c_void_p]),
    "ptk_host_register": (c_int, [c_void_p, c_uint64]),
    "ptk_host_unregister": (c_int, [c_void_p]),
    "ptk_tree64_create_from_points": (c_int, [c_void_p, c_uint64, c_uint32, c_uint64, c_int32, POINTER(c_void_p)]),
    "ptk_tree64_create_from_stream": (c_int, [c_void_p, c_uint64, c_uint32, c_void_p, c_uint64, c_int32,
                                              POINTER(c_void_p)]),
    "ptk_tree64_destroy": (None, [c_void_p]),
    "ptk_tree64_get_info": (c_int, [c_void_p, POINTER(_Info)]),
    "ptk_tree64_set_metric": (c_int, [c_void_p, c_int]),
    "ptk_tree64_serialize": (c_int, [c_void_p, c_void_p, c_uint64, POINTER(c_uint64)]),
    "ptk_tree64_serialize_topological": (c_int, [c_void_p, c_void_p, c_uint64, POINTER(c_uint64)]),
    "ptk_tree64_create_from_topological_stream": (c_int, [c_void_p, c_uint64, c_uint32, c_void_p, c_uint64, c_int32,
                                                          POINTER(c_void_p)]),
    "ptk_search64_knn": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_double, c_void_p]),
    "ptk_search64_knn_device": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_double, c_void_p, c_void_p]),
    "ptk_search64_radius": (c_int, [c_void_p, c_void_p, c_uint64, c_double, c_double, c_int, c_void_p,
                                    POINTER(c_void_p)]),
    "ptk_search64_box": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, POINTER(c_void_p)]),
    "ptk_forest_create": (c_int, [c_void_p, c_uint64, c_uint32, c_uint64, c_uint32, c_uint64, c_int32,
                                  POINTER(c_void_p)]),
    "ptk_forest_destroy": (None, [c_void_p]),
    "ptk_forest_get_rotations": (c_int, [c_void_p, c_void_p]),
    "ptk_forest_get_dropped": (c_int, [c_void_p, POINTER(c_uint64)]),
    "ptk_forest_search_knn": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_uint64, c_void_p]),
    "ptk_forest_search_knn_device": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_uint64, c_void_p,
                                             c_void_p]),
    "ptk_profile_enable": (c_int, [c_void_p, c_int]),
    "ptk_profile_get": (c_int, [c_void_p, POINTER(_Profile), c_int]),
    "ptk_profile_get_sized": (c_int, [c_void_p, c_void_p, c_uint64, c_int]),
    "ptk_debug_knn1_counts": (c_int, [c_void_p, POINTER(c_uint32)]),
    "ptk_debug_piles": (c_int, [c_void_p, POINTER(ctypes.c_uint64)]),
    "ptk_debug_batch_permutation": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p]),
    "ptk_debug_create_phases": (c_int, [c_void_p, POINTER(c_double)]),
    "ptk_debug_key_bits": (c_int, [c_void_p, c_uint64, POINTER(c_uint32)]),
    "ptk_debug_batch_order": (c_int, [c_void_p, POINTER(c_int)]),
    "ptk_debug_knn_coop_counts": (c_int, [c_void_p, POINTER(c_uint32)]),
    "ptk_debug_radius_coop_counts": (c_int, [c_void_p, POINTER(c_uint32)]),
    "ptk_tree64_debug_knn_coop_counts": (c_int, [c_void_p, POINTER(c_uint32)]),
    "ptk_debug_knn_cap": (c_int, [c_uint64, c_uint32, c_float, POINTER(c_uint32), POINTER(c_uint64)]),
    "ptk_multi_create_from_points": (c_int, [c_void_p, c_uint64, c_uint32, c_uint64, c_void_p, c_uint32,
                                             POINTER(c_void_p)]),
    "ptk_multi_create": (c_int, [c_void_p, c_void_p, c_uint32, POINTER(c_void_p)]),
    "ptk_multi_destroy": (None, [c_void_p]),
    "ptk_multi_device_count": (c_int, [c_void_p]),
    "ptk_multi_get_tree": (c_int, [c_void_p, c_uint32, POINTER(c_void_p)]),
    "ptk_multi_set_metric": (c_int, [c_void_p, c_int]),
    "ptk_multi_search_knn": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_float, c_void_p]),
    "ptk_multi_search_radius": (c_int, [c_void_p, c_void_p, c_uint64, c_float, c_float, c_int, c_void_p,
                                        POINTER(c_void_p)]),
    "ptk_multi_search_knn_device": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_float, c_void_p, c_void_p]),
}

#: Every symbol include/ptk.h declares; tests check the library exports them all.
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path() -> str:
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} is missing: build it with `python -m pico_tree_amd.build` "
                "(there is no CPU fallback)")
        # PyTorch bundles its own HIP runtime under the same SONAME as the system one
        # (libamdhip64.so.7).  Import it first so that libptk binds to the runtime that
        # torch tensors and streams live in, whatever the import order of the caller.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(_LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


PTK_ERR_UNSUPPORTED = -2
_host_loop_warned = False
_host_loop_allowed = False


def allow_host_loop(on: bool = True) -> None:
    """Off by default: a batched search the DEVICE refuses for a valid tree (``PTK_ERR_UNSUPPORTED`` -- a topological
    tree deeper than the device stack, a dimension beyond the LDS staging) raises :class:`PtkError`, like every other
    failure of the backend: the batched path has no CPU fallback.  ``allow_host_loop(True)`` lets the wrapper serve such a
    call with the library's host loop (``ptk_host_search_*``: the reference's own batch loop over the per-query search)
    after one ``RuntimeWarning``.  Process-wide."""
    global _host_loop_allowed
    _host_loop_allowed = bool(on)


def _warn_host_loop(why: str) -> None:
    """One warning per process when a call the device search refuses is served by the host loop instead."""
    global _host_loop_warned
    if not _host_loop_warned:
        _host_loop_warned = True
        import warnings
        warnings.warn("pico_tree_amd: the device search refused this call (" + why + "); it runs as a loop of "
                      "per-query host searches over the rows instead, as the reference's module does for every call "
                      "(further refusals are served the same way without this message)", RuntimeWarning, stacklevel=4)


def _check(status: int) -> None:
    if status != PTK_OK:
        raise PtkError(status, _load().ptk_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    return int(_load().ptk_device_count())


def warmup(device: int = -1) -> None:
    """Starts loading the library's device code for ``device`` (default: the current one) in the background, so that
    the first :class:`KdTree` for it does not wait for the load (``ptk_warmup``).  Optional."""
    _check(_load().ptk_warmup(int(device)))


class _PinnedBlock:
    """One block of page-locked host memory out of :class:`_PinnedPool`; numpy arrays made from it keep it alive
    (``__array_interface__``), and the block goes back to the pool when the last of them is gone."""

    def __init__(self, pool, ptr: int, capacity: int, nbytes: int):
        self._pool, self._ptr, self._capacity = pool, ptr, capacity
        self.__array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1", "version": 3}

    def __del__(self):
        pool, self._pool = self._pool, None
        if pool is not None:
            pool._give(self._ptr, self._capacity)


class _PinnedPool:
    """Result arrays of the host-buffer searches.  The reference's module returns a NEW array from every
    ``search_knn(pts, k)`` (def_kd_tree.cpp:73-82); a fresh 58 MB numpy array costs its first touch on every call
    (2-19 ms on the hosts of the GPU pool, profiles/r03_notes.txt item 24) and a staging copy on the way.  Here the
    rows land in page-locked blocks (``ptk_host_alloc``) that the device writes directly and that are handed out again
    once the array built on them has been garbage-collected.  ``PTK_PINNED_POOL_MB`` (default 2048) bounds what the
    pool holds, in use and free; beyond it -- or for small results -- plain numpy arrays are returned.  A request is
    served from the smallest idle block that holds it (within a factor of two: a workload whose batch sizes vary reuses
    its blocks instead of pinning one per size), and when the budget is spent the idle blocks of other sizes are freed
    before a pageable array is given out.  ``pico_tree_amd.trim_pinned_pool()`` frees every idle block."""

    MIN_BYTES = 1 << 20

    def __init__(self):
        import threading
        # (re-entrant: a block's __del__ may run -- cyclic garbage collection -- on the thread that holds the lock)
        self._lock = threading.RLock()
        self._free = {}   # capacity -> [ptr, ...]
        self._held = 0    # bytes allocated (in use + free)

    def _budget(self) -> int:
        return int(os.environ.get("PTK_PINNED_POOL_MB", "2048")) << 20

    def empty(self, shape, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes < self.MIN_BYTES:
            return np.empty(shape, dtype=dtype)
        capacity = (nbytes + (1 << 20) - 1) & ~((1 << 20) - 1)
        ptr = None
        evict = []
        with self._lock:
            # the smallest idle block that holds the request, if it is not more than twice as large
            # (snapshots of the dict: a block's __del__ may re-enter _give() on this thread -- garbage collection inside
            # the loop -- and add a capacity key: 'dictionary changed size during iteration')
            fits = sorted(c for c, stock in list(self._free.items()) if stock and capacity <= c <= 2 * capacity)
            if fits:
                capacity = fits[0]
                ptr = self._free[capacity].pop()
            else:
                if self._held + capacity > self._budget():  # idle blocks of other sizes make room first
                    for c in sorted(list(self._free), reverse=True):
                        while self._free[c] and self._held + capacity > self._budget():
                            evict.append(self._free[c].pop())
                            self._held -= c
                if self._held + capacity <= self._budget():
                    self._held += capacity
                    ptr = 0  # allocate outside the lock
        if evict:
            lib = _load()
            for p in evict:
                lib.ptk_host_free(p)
        if ptr is None:
            return np.empty(shape, dtype=dtype)
        if ptr == 0:
            out = c_void_p()
            if _load().ptk_host_alloc(capacity, byref(out)) != PTK_OK or not out.value:
                with self._lock:
                    self._held -= capacity
                return np.empty(shape, dtype=dtype)
            ptr = out.value
        block = _PinnedBlock(self, ptr, capacity, nbytes)
        return np.asarray(block).view(dtype).reshape(shape)

    def _give(self, ptr: int, capacity: int) -> None:
        try:
            with self._lock:
                self._free.setdefault(capacity, []).append(ptr)
        except Exception:  # (interpreter shutdown)
            pass

    def trim(self) -> None:
        """Frees the blocks that are not in use."""
        with self._lock:
            free, self._free = self._free, {}
            for capacity, stock in list(free.items()):
                self._held -= capacity * len(stock)
        lib = _load()
        for stock in free.values():
            for ptr in stock:
                lib.ptk_host_free(ptr)


_pinned_pool = _PinnedPool()


def set_test_knobs(**knobs) -> None:
    """Test hooks of libptk.so: ``set_test_knobs(knn_cap=4, knn_cap_min_nq=1)`` writes ``name=value`` pairs into the
    environment variable ``PTK_TEST_KNOBS`` (the library reads it on every call); ``name=None`` removes one,
    ``set_test_knobs()`` with no arguments removes all.  The hooks force a path a test wants to see taken (a cap on every
    batch, one of the two sorts, the hand-over list of a small batch ...); DESIGN.md section 11 lists them."""
    if not knobs:
        os.environ.pop("PTK_TEST_KNOBS", None)
        return
    have = dict(item.split("=", 1) for item in os.environ.get("PTK_TEST_KNOBS", "").split(",") if "=" in item)
    for name, value in knobs.items():
        if value is None:
            have.pop(name, None)
        else:
            have[name] = str(int(value))
    if have:
        os.environ["PTK_TEST_KNOBS"] = ",".join(f"{k}={v}" for k, v in have.items())
    else:
        os.environ.pop("PTK_TEST_KNOBS", None)


def trim_pinned_pool() -> None:
    """Frees the page-locked result blocks that no array uses any more (the pool keeps them for the next call)."""
    _pinned_pool.trim()


class registered:
    """``with pico_tree_amd.registered(arr): ...`` page-locks a numpy array the caller owns for the duration of the
    block (``ptk_host_register`` / ``ptk_host_unregister``): searches inside move it without a staging copy.  For arrays
    that live long -- the buffers of a stream of scans; registering costs more than one staged copy."""

    def __init__(self, arr: np.ndarray):
        if not isinstance(arr, np.ndarray) or not arr.flags["C_CONTIGUOUS"]:
            raise ValueError("a C-contiguous numpy array is required")
        self._arr = arr
        self._on = False

    def __enter__(self):
        _check(_load().ptk_host_register(self._arr.ctypes.data, self._arr.nbytes))
        self._on = True
        return self._arr

    def __exit__(self, *exc):
        if self._on:
            self._on = False
            _check(_load().ptk_host_unregister(self._arr.ctypes.data))
        return False


def empty_pinned(shape, dtype=np.float32) -> np.ndarray:
    """An uninitialised numpy array in page-locked host memory (``ptk_host_alloc``): query arrays kept in one are
    uploaded without staging.  Small arrays (< 1 MiB) are ordinary numpy arrays."""
    return _pinned_pool.empty(shape, dtype)


class Metric(enum.Enum):
    """Metrics of the reference binding (``def_kd_tree.cpp:14-17``) plus ``LNInf``, which the reference
    has in C++ only (``metric_lninf``, metric.hpp:157-186).  ``L2Squared`` takes the tuned kernels; the
    others run the generic kernels with the metric swapped in."""
    L1 = 1
    L2Squared = 2
    LPInf = 3
    LNInf = 4
    #: the reference's topological metrics (C++ only there, metric.hpp:186-257): points on the circle
    #: [0, 1] / 0 ~ 1 (1-D), and planar poses x, y, angle (3-D).  float32; knn and radius searches.
    SO2 = 5
    SE2Squared = 6


_PTK_METRIC = {Metric.L2Squared: 0, Metric.L1: 1, Metric.LPInf: 2, Metric.LNInf: 3, Metric.SO2: 4,
               Metric.SE2Squared: 5}  # PTK_METRIC_* of ptk.h


class _LibraryBuffer:
    """Owner of a result buffer malloc'ed by libptk: frees it (``ptk_free``) when the last numpy view
    of it is gone."""

    def __init__(self, lib, ptr):
        self._lib, self._ptr = lib, ptr

    def __del__(self):
        try:
            self._lib.ptk_free(self._ptr)
        except Exception:
            pass


def _adopt(lib, ptr: c_void_p, count: int, dtype) -> np.ndarray:
    """The library's ragged result as a numpy array WITHOUT copying it (a radius batch of BASELINE
    config 3 is 6 GB of rows): the array keeps the buffer alive through a ``_LibraryBuffer``."""
    dtype = np.dtype(dtype)
    if count == 0 or not ptr.value:
        lib.ptk_free(ptr)
        return np.empty(0, dtype=dtype)
    raw = (ctypes.c_char * (count * dtype.itemsize)).from_address(ptr.value)
    raw._owner = _LibraryBuffer(lib, ptr)  # the ctypes object is the base of the array below
    return np.frombuffer(raw, dtype=dtype, count=count)


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class DArray:
    """Ragged result of a radius (or box) search: a sequence of numpy arrays.

    Same role and constructor as the reference's ``DArray``
    (``_pyco_tree/darray.hpp:31-288``, ``def_darray.cpp``): ``DArray(dtype)`` makes an
    empty one to be filled by ``search_radius(..., nns)``; it is a sequence (``len``,
    indexing incl. negative indices and slices, iteration, truthiness).  Stored flat:
    row ``i`` is ``flat[offsets[i]:offsets[i + 1]]`` (a view, no copy).
    """

    def __init__(self, dtype=None, flat=None, *, offsets=None):
        if isinstance(dtype, np.ndarray) and isinstance(flat, np.ndarray):
            offsets, dtype = dtype, None  # DArray(offsets, flat): the internal form
        if flat is None:
            dt = np.dtype(dtype if dtype is not None else NEIGHBOR)
            if dt != NEIGHBOR and dt != NEIGHBOR64 and dt != np.dtype(np.int32):
                raise ValueError("unexpected dtype for DArray")
            flat = np.empty(0, dtype=dt)
            offsets = np.zeros(1, dtype=np.uint64)
        self.offsets = offsets
        self.flat = flat

    @property
    def dtype(self):
        return self.flat.dtype

    def __len__(self) -> int:
        return len(self.offsets) - 1

    def __bool__(self) -> bool:
        return len(self) > 0

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            rows = [self[j] for j in range(*i.indices(n))]
            off = np.zeros(len(rows) + 1, dtype=np.uint64)
            if rows:
                off[1:] = np.cumsum([len(r) for r in rows])
            flat = np.concatenate(rows) if rows else np.empty(0, dtype=self.dtype)
            return DArray(off, flat.astype(self.dtype, copy=False))
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return self.flat[int(self.offsets[i]):int(self.offsets[i + 1])]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def _assign(self, offsets: np.ndarray, flat: np.ndarray) -> None:
        """Takes a new result; storage is kept when the total size is unchanged (the
        reference re-uses each row's memory when its size allows, darray.hpp)."""
        if flat.dtype != self.flat.dtype:
            raise ValueError("unexpected dtype_neighbor for data")
        if len(flat) == len(self.flat) and len(flat) > 0:
            np.copyto(self.flat, flat)
        else:
            self.flat = flat
        self.offsets = offsets


class DeviceNeighbors:
    """k-NN result resident on the device: ``raw`` is int32 ``(nq, k, 2)`` -- (index, bits of the
    float32 distance) -- for a float32 tree and int64 ``(nq, k, 2)`` -- (index, bits of the float64
    distance) -- for a float64 tree (the 16-byte ``neighbor<int, double>`` record)."""

    def __init__(self, raw):
        self.raw = raw

    @property
    def index(self):
        return self.raw[..., 0]

    @property
    def distance(self):
        import torch
        return self.raw[..., 1].view(torch.float64 if self.raw.dtype == torch.int64 else torch.float32)

    def numpy(self) -> np.ndarray:
        """Host copy with the :data:`NEIGHBOR` (or :data:`NEIGHBOR64`) dtype, shape ``(nq,)`` or ``(nq, k)``."""
        a = self.raw.cpu().numpy()
        out = np.ascontiguousarray(a).view(NEIGHBOR64 if a.dtype == np.int64 else NEIGHBOR)[..., 0]
        return out[:, 0] if out.shape[1] == 1 else out


class KdTree:
    """A kd-tree over float32 or float64 points, searched on the MI355X.

    ``KdTree(pts, Metric.L2Squared, max_leaf_size)`` as in the reference.  ``pts``
    is ``(npts, sdim)`` C-contiguous (an F-contiguous ``(sdim, npts)`` array is the
    same memory and accepted like the reference does); its dtype selects the
    instantiation like the reference's dispatch does (_pyco_tree/kd_tree.hpp:383-445):
    float32 -> the tuned kernels (``ptk_*``), float64 -> the double-precision
    kernels (``ptk_tree64_*`` / ``ptk_search64_*``, results ``NEIGHBOR64``).  The tree
    is built on the host with the sliding-midpoint rule and uploaded once.
    """

    def __init__(self, pts, metric: Metric = Metric.L2Squared, max_leaf_size: int = 10,
                 device: int | None = None, _stream: bytes | None = None):
        if not isinstance(metric, Metric):
            raise TypeError("metric must be a pico_tree_amd.Metric")
        self._metric = metric
        if isinstance(pts, np.ndarray) and pts.dtype == np.float64:
            self._dtype, self._neighbor, self._f64 = np.dtype(np.float64), NEIGHBOR64, True
        else:
            self._dtype, self._neighbor, self._f64 = np.dtype(np.float32), NEIGHBOR, False
        self._real = np.float64 if self._f64 else np.float32
        self._given = pts  # in the caller's own layout (row- or column-major), for __array__
        pts = self._as_matrix(pts, None, "pts", self._dtype)
        if int(max_leaf_size) <= 0:
            raise ValueError("max_leaf_size must be positive")
        self._pts = pts  # keep alive, like py::keep_alive<1, 2>
        self._npts, self._sdim = pts.shape
        self._max_leaf_size = int(max_leaf_size)
        lib = _load()
        handle = c_void_p()
        dev = PTK_DEVICE_CURRENT if device is None else int(device)
        self._fn = (lambda name: getattr(lib, name.replace("ptk_tree_", "ptk_tree64_").replace("ptk_search_", "ptk_search64_"))) \
            if self._f64 else (lambda name: getattr(lib, name))
        if _stream is None:
            _check(self._fn("ptk_tree_create_from_points")(pts.ctypes.data, self._npts, self._sdim,
                                                           self._max_leaf_size, dev, byref(handle)))
        else:  # load_kd_tree: the tree comes from a saved stream
            buf = ctypes.create_string_buffer(_stream, len(_stream))
            # (a tree over a topological space is written with four bounds per branch, kd_tree_node.hpp:52-67)
            entry = "ptk_tree_create_from_topological_stream" if self._topological() else "ptk_tree_create_from_stream"
            _check(self._fn(entry)(pts.ctypes.data, self._npts, self._sdim, buf, len(_stream), dev, byref(handle)))
        self._h = handle
        if metric is not Metric.L2Squared:
            _check(self._fn("ptk_tree_set_metric")(handle, _PTK_METRIC[metric]))

    def __array__(self, dtype=None, copy=None):
        """The tree as a read-only view of its points, shaped as they were given -- ``(npts, sdim)`` for
        row-major input, ``(sdim, npts)`` for column-major input: the reference exposes exactly this
        through the buffer protocol (``def_kd_tree.cpp:30-33``, ``_pyco_tree/kd_tree.hpp:292-315``) to
        show how the data is interpreted.  ``np.asarray(tree)`` does not copy."""
        v = self._given.view()
        v.flags.writeable = False
        if dtype is not None and np.dtype(dtype) != v.dtype:
            return v.astype(dtype)
        return v.copy() if copy else v

    @property
    def metric_string(self) -> str:  # core.hpp:24-38
        return self._metric.name

    def _topological(self) -> bool:
        return self._metric.name in ("SO2", "SE2Squared")

    def _serialize(self) -> bytes:
        entry = "ptk_tree_serialize_topological" if self._topological() else "ptk_tree_serialize"
        size = c_uint64()
        _check(self._fn(entry)(self._h, None, 0, byref(size)))
        buf = ctypes.create_string_buffer(size.value)
        _check(self._fn(entry)(self._h, buf, size.value, byref(size)))
        return buf.raw[:size.value]

    # -- helpers ---------------------------------------------------------------
    @staticmethod
    def _as_matrix(a, sdim, what, dtype=np.float32):
        """Validates like py_array_map.hpp:37-62 and returns a C-order view."""
        if not isinstance(a, np.ndarray):
            raise ValueError(f"{what} must be a numpy array")
        if a.dtype != dtype:
            raise ValueError("unexpected dtype_scalar for data")
        if a.ndim != 2:
            raise ValueError(f"{what} must have 2 dimensions")
        if a.flags.c_contiguous:
            m = a
        elif a.flags.f_contiguous:
            m = a.T  # (sdim, npts) column-major is (npts, sdim) row-major memory
        else:
            raise ValueError(f"{what} must be contiguous")
        if sdim is not None and m.shape[1] != sdim:
            raise ValueError(f"{what} has spatial dimension {m.shape[1]}, tree has {sdim}")
        return m

    def close(self) -> None:
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._fn("ptk_tree_destroy")(h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self) -> str:  # _pyco_tree/kd_tree.hpp:321-326
        return f"KdTree(metric={self._metric.name}, dtype={self._dtype.name}, sdim={self._sdim}, npts={self._npts})"

    # -- properties (names of the reference binding) ----------------------------------
    @property
    def npts(self) -> int:
        return self._npts

    @property
    def sdim(self) -> int:
        return self._sdim

    @property
    def dtype_index(self):
        return np.dtype(np.int32)

    @property
    def dtype_scalar(self):
        return self._dtype

    @property
    def dtype_neighbor(self):
        return self._neighbor

    def metric(self, scalar: float) -> float:
        """The metric's one-dimensional form in the tree's scalar type: ``x * x`` for L2Squared, ``|x|``
        for L1 and LPInf (metric.hpp:95-98, :120-123, :147-150)."""
        x = self._real(scalar)
        return float(x * x) if self._metric in (Metric.L2Squared, Metric.SE2Squared) else float(abs(x))

    def info(self) -> dict:
        inf = _Info()
        _check(self._fn("ptk_tree_get_info")(self._h, byref(inf)))
        return {name: getattr(inf, name) for name, _ in _Info._fields_}

    def _float32_only(self, what):
        if self._f64:
            raise PtkError(-2, f"{what} is available for float32 trees only")

    def flat(self):
        """(nodes uint32[n_nodes, 4], indices int32[npts], root_min, root_max)."""
        self._float32_only("flat()")
        inf = self.info()
        nodes = np.empty((inf["n_nodes"], 4), dtype=np.uint32)
        indices = np.empty(self._npts, dtype=np.int32)
        rmin = np.empty(self._sdim, dtype=np.float32)
        rmax = np.empty(self._sdim, dtype=np.float32)
        _check(_load().ptk_tree_get_flat(self._h, nodes.ctypes.data, indices.ctypes.data,
                                         rmin.ctypes.data, rmax.ctypes.data))
        return nodes, indices, rmin, rmax

    def set_reorder(self, mode: int) -> None:
        if self._f64:
            return  # the double-precision kernels take the batch in the caller's order
        _check(_load().ptk_tree_set_reorder(self._h, int(mode)))

    def profile(self, enable: bool | None = None, reset: bool = False) -> dict:
        self._float32_only("profile()")
        lib = _load()
        if enable is not None:
            _check(lib.ptk_profile_enable(self._h, int(enable)))
        p = _Profile()
        _check(lib.ptk_profile_get_sized(self._h, byref(p), ctypes.sizeof(p), int(reset)))
        return {name: getattr(p, name) for name, _ in _Profile._fields_}

    def create_phases(self) -> dict:
        """Where the creation of the handle went, in seconds (``ptk_debug_create_phases``)."""
        self._float32_only("create_phases()")
        ms = (c_double * 3)()
        _check(_load().ptk_debug_create_phases(self._h, ms))
        return {"host_build_s": ms[0] / 1e3, "encode_s": ms[1] / 1e3, "upload_s": ms[2] / 1e3}

    def knn1_counts(self) -> dict:
        """Counters of the last two-phase k = 1 search (``ptk_debug_knn1_counts``)."""
        self._float32_only("knn1_counts()")
        c = (c_uint32 * 4)()
        _check(_load().ptk_debug_knn1_counts(self._h, c))
        return {"phase2": int(c[0]), "cooperative": int(c[1]), "redone": int(c[2]), "dealt": int(c[3])}

    def knn_coop_counts(self) -> dict:
        """After a search with 1 < k <= 56: queries the general kernel handed to the cooperative search, queries that
        search sent to the redo list and why (``ptk_debug_knn_coop_counts``)."""
        c = (c_uint32 * 7)()
        if self._f64:
            _check(_load().ptk_tree64_debug_knn_coop_counts(self._h, c))
        else:
            _check(_load().ptk_debug_knn_coop_counts(self._h, c))
        return {"cooperative": int(c[0]), "redone": int(c[1]), "pool": int(c[2]), "ties": int(c[3]), "box": int(c[4]),
                "range": int(c[5]), "tie_sweeps": int(c[6])}

    def radius_coop_counts(self) -> dict:
        """After a radius count pass on the device (``search_radius_device`` / the count half of ``search_radius``):
        queries the list pass handed to a wavefront, rows recounted from the root, leaf entries kept for the fill pass
        (``ptk_debug_radius_coop_counts``)."""
        self._float32_only("radius_coop_counts()")
        c = (c_uint32 * 3)()
        _check(_load().ptk_debug_radius_coop_counts(self._h, c))
        return {"cooperative": int(c[0]), "recounted": int(c[1]), "entries": int(c[2])}

    def piles(self) -> dict:
        """Subtrees of coincident points of the device replica (``ptk_debug_piles``): how many, the points they hold,
        the depth of what a k = 1 search of the default metric traverses."""
        self._float32_only("piles()")
        c = (ctypes.c_uint64 * 3)()
        _check(_load().ptk_debug_piles(self._h, c))
        return {"piles": int(c[0]), "points": int(c[1]), "knn1_depth": int(c[2])}

    def batch_permutation(self, q):
        """The order the rows of a device batch would be searched in (``ptk_debug_batch_permutation``): a uint32 torch
        tensor, entry i = the row searched i-th."""
        import torch

        self._float32_only("batch_permutation()")
        perm = torch.empty((q.shape[0],), dtype=torch.int32, device=q.device)
        _check(_load().ptk_debug_batch_permutation(self._h, q.data_ptr(), q.shape[0], perm.data_ptr()))
        return perm

    def batch_order(self) -> int:
        """What the last search did with the order of its batch: 0 as it came, 1 sorted on the device, 2 found
        coherent and left alone (``ptk_debug_batch_order``)."""
        self._float32_only("batch_order()")
        how = c_int(0)
        _check(_load().ptk_debug_batch_order(self._h, byref(how)))
        return int(how.value)

    def key_bits(self, nq: int) -> tuple:
        """Bits of the Morton key per axis for a batch of ``nq`` queries (``ptk_debug_key_bits``)."""
        self._float32_only("key_bits()")
        b = (c_uint32 * 3)()
        _check(_load().ptk_debug_key_bits(self._h, int(nq), b))
        return int(b[0]), int(b[1]), int(b[2])

    # -- k nearest neighbours ------------------------------------------------------------
    def search_knn(self, pts, k: int, *args):
        """``search_knn(pts, k[, e][, nns])`` -- the four reference overloads.

        Host arrays return (or fill) a numpy array of :data:`NEIGHBOR`; a torch
        CUDA tensor returns a :class:`DeviceNeighbors` (``nns`` may be a
        preallocated int32 ``(nq, k, 2)`` tensor).
        """
        e, nns = self._split_optional(args)
        k = int(k)
        if _is_torch(pts):
            return self._search_knn_device(pts, k, e, nns)
        q = self._as_matrix(pts, self._sdim, "pts", self._dtype)
        nq = q.shape[0]
        shape = (nq,) if k == 1 else ((nq, k) if pts.flags.c_contiguous else (k, nq))
        NB = self._neighbor
        if nns is None:
            # (float64 records carry 4 bytes of padding, zeroed; float32 rows land in a page-locked block of the pool)
            nns = np.zeros(shape, dtype=NB) if self._f64 else _pinned_pool.empty(shape, NB)
        elif not isinstance(nns, np.ndarray) or nns.dtype != NB:
            raise ValueError("unexpected dtype_neighbor for data")
        elif nns.size != nq * k or not nns.flags.c_contiguous:
            # Resized like ensure_size() of the reference (kd_tree.hpp:362-378).
            try:
                nns.resize(shape, refcheck=False)
            except ValueError:
                nns = np.empty(shape, dtype=NB)
        search = self._fn("ptk_search_knn")
        if k > 1 and not pts.flags.c_contiguous:
            # Column-major callers get the transposed (k, npts) layout of the reference
            # (kd_tree.hpp:362-378): row i of the search is column i of the output.
            tmp = np.empty((nq, k), dtype=NB)
            self._served(search(self._h, q.ctypes.data, nq, k, self._real(e), tmp.ctypes.data),
                         lambda lib: lib.ptk_host_search_knn(self._h, self._pts.ctypes.data, q.ctypes.data, nq, k,
                                                             np.float32(e), tmp.ctypes.data))
            nns.reshape(-1)[:] = tmp.reshape(-1)
            return nns
        self._served(search(self._h, q.ctypes.data, nq, k, self._real(e), nns.ctypes.data),
                     lambda lib: lib.ptk_host_search_knn(self._h, self._pts.ctypes.data, q.ctypes.data, nq, k,
                                                         np.float32(e), nns.ctypes.data))
        return nns

    def _served(self, status: int, host_loop) -> None:
        """``_check``: every failure raises.  Only after ``allow_host_loop(True)`` is a search the DEVICE refuses for a
        valid tree (PTK_ERR_UNSUPPORTED) served by the host loop of the library (``ptk_host_search_*``: the reference's
        own batch loop) with one warning; a missing device or a HIP error raises regardless."""
        if status == PTK_ERR_UNSUPPORTED and not self._f64 and _host_loop_allowed:
            lib = _load()
            _warn_host_loop(lib.ptk_last_error().decode("utf-8", "replace"))
            status = host_loop(lib)
        _check(status)

    def _search_knn_device(self, q, k, e, out):
        import torch
        tq, traw = (torch.float64, torch.int64) if self._f64 else (torch.float32, torch.int32)
        if q.dtype != tq or q.dim() != 2 or q.shape[1] != self._sdim:
            raise ValueError(f"queries must be a {self._dtype.name} (nq, sdim) tensor")
        if not q.is_cuda or not q.is_contiguous():
            raise ValueError("queries must be a contiguous CUDA tensor")
        nq = q.shape[0]
        if out is None:
            # float64: the records carry 4 bytes of padding, zeroed so that index reads as an int64
            out = (torch.zeros if self._f64 else torch.empty)((nq, k, 2), dtype=traw, device=q.device)
        elif isinstance(out, DeviceNeighbors):
            out = out.raw
        if out.dtype != traw or tuple(out.shape) != (nq, k, 2) or not out.is_contiguous():
            raise ValueError(f"nns must be a contiguous {traw} (nq, k, 2) tensor")
        stream = torch.cuda.current_stream(q.device).cuda_stream
        _check(self._fn("ptk_search_knn_device")(self._h, q.data_ptr(), nq, k, self._real(e),
                                                 out.data_ptr(), stream))
        return DeviceNeighbors(out)

    # -- radius ---------------------------------------------------------------------------
    def search_radius(self, pts, radius: float, *args, sort: bool = False):
        """``search_radius(pts, radius[, e][, nns], sort=False)`` -> :class:`DArray`."""
        e, nns, sort = self._split_optional_radius(args, sort)
        q = self._as_matrix(pts, self._sdim, "pts", self._dtype)
        nq = q.shape[0]
        if nns is not None and (not isinstance(nns, DArray) or nns.dtype != self._neighbor):
            raise ValueError("unexpected dtype_neighbor for data")
        offsets = np.zeros(nq + 1, dtype=np.uint64)
        rows = c_void_p()
        lib = _load()
        self._served(self._fn("ptk_search_radius")(self._h, q.ctypes.data, nq, self._real(radius),
                                                   self._real(e), int(bool(sort)), offsets.ctypes.data, byref(rows)),
                     lambda lib: lib.ptk_host_search_radius(self._h, self._pts.ctypes.data, q.ctypes.data, nq,
                                                            np.float32(radius), np.float32(e), int(bool(sort)),
                                                            offsets.ctypes.data, byref(rows)))
        flat = _adopt(lib, rows, int(offsets[-1]), self._neighbor)
        if nns is None:
            return DArray(offsets, flat)
        nns._assign(offsets, flat)
        return nns

    # -- box ------------------------------------------------------------------------------
    def search_box(self, boxes, nns=None):
        """``search_box(boxes[, nns])`` -> :class:`DArray` of int32 index arrays.

        ``boxes`` is ``(2 * nbox, sdim)``: rows ``2 i`` and ``2 i + 1`` are the min and max corner
        of box ``i`` (``_pyco_tree/kd_tree.hpp:245-268``).  Row ``i`` of the result lists the points
        inside the closed box in the reference's traversal order."""
        b = self._as_matrix(boxes, self._sdim, "boxes", self._dtype)
        if b.shape[0] % 2 != 0:
            raise ValueError("query min and max don't have equal size")
        nb = b.shape[0] // 2
        mins = np.ascontiguousarray(b[0::2])
        maxs = np.ascontiguousarray(b[1::2])
        if nns is not None and (not isinstance(nns, DArray) or nns.dtype != np.dtype(np.int32)):
            raise ValueError("unexpected dtype_index for data")
        offsets = np.zeros(nb + 1, dtype=np.uint64)
        rows = c_void_p()
        lib = _load()
        self._served(self._fn("ptk_search_box")(self._h, mins.ctypes.data, maxs.ctypes.data, nb, offsets.ctypes.data,
                                                byref(rows)),
                     lambda lib: lib.ptk_host_search_box(self._h, self._pts.ctypes.data, mins.ctypes.data,
                                                         maxs.ctypes.data, nb, offsets.ctypes.data, byref(rows)))
        flat = _adopt(lib, rows, int(offsets[-1]), np.int32)
        if nns is None:
            return DArray(offsets, flat)
        nns._assign(offsets, flat)
        return nns

    def search_box_device(self, mins, maxs):
        """Device form of :meth:`search_box`: ``mins`` / ``maxs`` are float32 ``(nbox, sdim)`` CUDA tensors; returns
        (offsets int64 tensor [nbox + 1], indices int32 tensor [total]) on the current torch stream."""
        import torch
        self._float32_only("search_box_device()")
        for a in (mins, maxs):
            if a.dtype != torch.float32 or a.dim() != 2 or a.shape[1] != self._sdim or not a.is_cuda \
                    or not a.is_contiguous():
                raise ValueError("mins / maxs must be contiguous float32 (nbox, sdim) CUDA tensors")
        if mins.shape != maxs.shape:
            raise ValueError("query min and max don't have equal size")
        nb = mins.shape[0]
        lib = _load()
        stream = torch.cuda.current_stream(mins.device).cuda_stream
        counts = torch.zeros(nb + 1, dtype=torch.int64, device=mins.device)
        _check(lib.ptk_search_box_count_device(self._h, mins.data_ptr(), maxs.data_ptr(), nb, counts.data_ptr(), stream))
        offsets = torch.zeros(nb + 1, dtype=torch.int64, device=mins.device)
        offsets[1:] = torch.cumsum(counts[:nb], 0)
        total = int(offsets[-1].item())
        out = torch.empty(max(total, 1), dtype=torch.int32, device=mins.device)
        _check(lib.ptk_search_box_fill_device(self._h, mins.data_ptr(), maxs.data_ptr(), nb, offsets.data_ptr(),
                                              out.data_ptr(), stream))
        return offsets, out[:total]

    def search_radius_device(self, q, radius: float, e: float = 1.0, sort: bool = False):
        """Device form: returns (offsets int64 tensor [nq + 1], raw int32 tensor [total, 2])."""
        import torch
        self._float32_only("search_radius_device()")
        if q.dtype != torch.float32 or q.dim() != 2 or q.shape[1] != self._sdim:
            raise ValueError("queries must be a float32 (nq, sdim) tensor")
        if not q.is_cuda or not q.is_contiguous():
            raise ValueError("queries must be a contiguous CUDA tensor")
        nq = q.shape[0]
        lib = _load()
        stream = torch.cuda.current_stream(q.device).cuda_stream
        counts = torch.zeros(nq + 1, dtype=torch.int64, device=q.device)
        _check(lib.ptk_search_radius_count_device(self._h, q.data_ptr(), nq, np.float32(radius),
                                                  np.float32(e), counts.data_ptr(), stream))
        offsets = torch.zeros(nq + 1, dtype=torch.int64, device=q.device)
        offsets[1:] = torch.cumsum(counts[:nq], 0)
        total = int(offsets[-1].item())
        out = torch.empty((max(total, 1), 2), dtype=torch.int32, device=q.device)
        _check(lib.ptk_search_radius_fill_device(self._h, q.data_ptr(), nq, np.float32(radius),
                                                 np.float32(e), offsets.data_ptr(), out.data_ptr(),
                                                 int(bool(sort)), stream))
        return offsets, out[:total]

    # -- argument plumbing for the overload sets ------------------------------------------------
    @staticmethod
    def _split_optional(args):
        e, nns = 1.0, None
        for a in args:
            if isinstance(a, (int, float, np.floating)) and not isinstance(a, bool):
                e = float(a)
            else:
                nns = a
        if len(args) > 2:
            raise TypeError("search_knn(pts, k[, e][, nns])")
        if not e > 0:
            raise ValueError("e must be positive")
        return e, nns

    @staticmethod
    def _split_optional_radius(args, sort):
        e, nns = 1.0, None
        for a in args:
            if isinstance(a, bool):
                sort = a
            elif isinstance(a, (int, float, np.floating)):
                e = float(a)
            else:
                nns = a
        if len(args) > 3:
            raise TypeError("search_radius(pts, radius[, e][, nns][, sort])")
        if not e > 0:
            raise ValueError("e must be positive")
        return e, nns, sort


class MultiKdTree:
    """One kd-tree replicated on several GPUs of this node, single process (``ptk_multi_*``).

    A batch is cut into ``len(devices)`` contiguous row ranges, range ``r`` searched on
    ``devices[r]``; rows come back in the caller's order and are bit-identical to a single-device
    :class:`KdTree`.  Host arrays move range by range (no inter-GPU traffic); with torch tensors on
    ``devices[0]`` the ranges and the (index, distance) rows travel over xGMI through RCCL."""

    def __init__(self, pts, max_leaf_size: int = 10, devices=None):
        p = np.ascontiguousarray(pts, dtype=np.float32)
        if p.ndim != 2:
            raise ValueError("pts must be (n, sdim)")
        if devices is None:
            devices = list(range(device_count()))
        self._devices = np.ascontiguousarray(devices, dtype=np.int32)
        self._sdim = p.shape[1]
        self._h = c_void_p()
        _check(_load().ptk_multi_create_from_points(p.ctypes.data, p.shape[0], p.shape[1], int(max_leaf_size),
                                                     self._devices.ctypes.data, len(self._devices), byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            _load().ptk_multi_destroy(self._h)
            self._h = None

    @property
    def devices(self):
        return [int(d) for d in self._devices]

    def _queries(self, q):
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self._sdim:
            raise ValueError("pts must be (nq, sdim)")
        return q

    def search_knn(self, pts, k: int, e: float | None = None):
        """Host arrays: ``(nq,)`` (k = 1) or ``(nq, k)`` :data:`NEIGHBOR` records; a torch tensor on
        ``devices[0]``: :class:`DeviceNeighbors` (asynchronous on the current torch stream)."""
        k = int(k)
        if _is_torch(pts):
            import torch

            if pts.dtype != torch.float32 or pts.dim() != 2 or pts.shape[1] != self._sdim or not pts.is_contiguous():
                raise ValueError("pts must be a contiguous float32 (nq, sdim) tensor")
            if pts.device.index != int(self._devices[0]):
                raise ValueError("pts must live on devices[0]")
            out = torch.empty((pts.shape[0], k, 2), dtype=torch.int32, device=pts.device)
            stream = torch.cuda.current_stream(pts.device).cuda_stream
            _check(_load().ptk_multi_search_knn_device(self._h, pts.data_ptr(), pts.shape[0], k,
                                                       np.float32(1.0 if e is None else e), out.data_ptr(), stream))
            return DeviceNeighbors(out)
        q = self._queries(pts)
        out = np.empty((q.shape[0],) if k == 1 else (q.shape[0], k), dtype=NEIGHBOR)
        _check(_load().ptk_multi_search_knn(self._h, q.ctypes.data, q.shape[0], k, np.float32(1.0 if e is None else e),
                                            out.ctypes.data))
        return out

    def search_radius(self, pts, radius: float, e: float | None = None, sort: bool = False) -> "DArray":
        q = self._queries(pts)
        offsets = np.zeros(q.shape[0] + 1, dtype=np.uint64)
        rows = c_void_p()
        lib = _load()
        _check(lib.ptk_multi_search_radius(self._h, q.ctypes.data, q.shape[0], np.float32(radius),
                                           np.float32(1.0 if e is None else e), int(bool(sort)), offsets.ctypes.data,
                                           byref(rows)))
        return DArray(offsets, _adopt(lib, rows, int(offsets[-1]), NEIGHBOR))


class KdForest:
    """Randomised kd-forest for approximate nearest neighbours in high dimensions, searched on
    the MI355X (one query per wavefront; ``pico_tree_amd/csrc/ptk_forest.hpp``).

    Mirrors ``pico_tree::kd_forest`` of the reference's pico_understory
    (``/root/reference/examples/pico_understory/pico_understory/kd_forest.hpp:42-85``):
    ``KdForest(pts, max_leaf_size, forest_size)``, ``search_nn(pts, max_leaves_visited)`` and, for
    k > 1, ``search_knn(pts, k, max_leaves_visited)`` (the reference's generic
    ``search_nearest`` with a k-list visitor, de-duplicated by index here).  ``seed`` fixes the
    Householder reflections, which the reference draws from ``std::random_device``.
    """

    def __init__(self, pts, max_leaf_size: int, forest_size: int, seed: int = 0, device: int | None = None):
        pts = KdTree._as_matrix(pts, None, "pts")
        if int(max_leaf_size) <= 0 or int(forest_size) <= 0:
            raise ValueError("max_leaf_size and forest_size must be positive")
        self._pts = pts
        self._npts, self._sdim = pts.shape
        self._forest_size = int(forest_size)
        handle = c_void_p()
        dev = PTK_DEVICE_CURRENT if device is None else int(device)
        _check(_load().ptk_forest_create(pts.ctypes.data, self._npts, self._sdim, int(max_leaf_size),
                                         self._forest_size, int(seed), dev, byref(handle)))
        self._h = handle

    @property
    def npts(self) -> int:
        return self._npts

    @property
    def sdim(self) -> int:
        return self._sdim

    @property
    def rotations(self) -> np.ndarray:
        """The unit reflection vectors, ``(forest_size, sdim)``."""
        out = np.empty((self._forest_size, self._sdim), dtype=np.float32)
        _check(_load().ptk_forest_get_rotations(self._h, out.ctypes.data))
        return out

    def search_knn(self, pts, k: int, max_leaves_visited: int):
        """Host array -> ``(nq, k)`` :data:`NEIGHBOR` array; torch CUDA tensor -> :class:`DeviceNeighbors`.
        Rows are ascending; slots beyond the distinct points found hold ``(-1, FLT_MAX)``."""
        k = int(k)
        if _is_torch(pts):
            import torch
            if pts.dtype != torch.float32 or pts.dim() != 2 or pts.shape[1] != self._sdim \
                    or not pts.is_cuda or not pts.is_contiguous():
                raise ValueError("queries must be a contiguous float32 (nq, sdim) CUDA tensor")
            out = torch.empty((pts.shape[0], k, 2), dtype=torch.int32, device=pts.device)
            stream = torch.cuda.current_stream(pts.device).cuda_stream
            _check(_load().ptk_forest_search_knn_device(self._h, pts.data_ptr(), pts.shape[0], k,
                                                        int(max_leaves_visited), out.data_ptr(), stream))
            return DeviceNeighbors(out)
        q = KdTree._as_matrix(pts, self._sdim, "pts")
        out = np.empty((q.shape[0], k), dtype=NEIGHBOR)
        _check(_load().ptk_forest_search_knn(self._h, q.ctypes.data, q.shape[0], k, int(max_leaves_visited),
                                             out.ctypes.data))
        return out

    @property
    def dropped(self) -> int:
        """Queue entries dropped so far because a per-tree queue was full (0 in normal use)."""
        v = c_uint64()
        _check(_load().ptk_forest_get_dropped(self._h, byref(v)))
        return int(v.value)

    def search_nn(self, pts, max_leaves_visited: int):
        """``(nq,)`` nearest neighbours (kd_forest.hpp:78-85)."""
        res = self.search_knn(pts, 1, max_leaves_visited)
        return res if isinstance(res, DeviceNeighbors) else res[:, 0]

    def close(self) -> None:
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            _load().ptk_forest_destroy(h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- file I/O (the reference's PKD container, _pyco_tree/kd_tree.hpp:546-614) -------------------
_PKD_SIGNATURE = b"\x89PKD"
_PKD_VERSION = 1


def save_kd_tree(tree: KdTree, filename: str) -> None:
    """``pico_tree.save_kd_tree``: PKD header (signature, version, metric string) followed by the
    tree in the reference's kd_tree::save format.  The points are not stored."""
    metric = tree.metric_string.encode("ascii")
    with open(filename, "wb") as f:
        f.write(_PKD_SIGNATURE)
        f.write(np.uint32(_PKD_VERSION).tobytes())
        f.write(np.uint64(len(metric)).tobytes())
        f.write(metric)
        f.write(tree._serialize())


def load_kd_tree(pts, filename: str, device: int | None = None) -> KdTree:
    """``pico_tree.load_kd_tree``: rebuilds a :class:`KdTree` over ``pts`` from a file written by
    :func:`save_kd_tree` or by the reference's own ``save_kd_tree`` (any of its metrics; the dtype of
    ``pts`` -- float32 or float64 -- must be the one the tree was built over)."""
    with open(filename, "rb") as f:
        data = f.read()
    if data[:4] != _PKD_SIGNATURE:
        raise RuntimeError("unexpected header signature")
    if int(np.frombuffer(data, dtype=np.uint32, count=1, offset=4)[0]) != _PKD_VERSION:
        raise RuntimeError("unsupported header version")
    n = int(np.frombuffer(data, dtype=np.uint64, count=1, offset=8)[0])
    metric = data[16:16 + n].decode("ascii", "replace")
    if metric not in Metric.__members__:  # kd_tree.hpp:588-598
        raise RuntimeError("unexpected metric string")
    return KdTree(pts, Metric[metric], 1, device=device, _stream=data[16 + n:])
