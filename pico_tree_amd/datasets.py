"""Seeded synthetic point clouds for the BASELINE.json configurations.

The reference benchmark (``/root/reference/docs/benchmark.md:11-13``) uses the
Bremen LiDAR scans, which are not redistributable here; BASELINE.md section 3
defines the stand-ins generated below.  Everything is derived from a
counter-based SplitMix64 stream evaluated with integer numpy ops only, so the
raw 24-bit uniforms are bit-reproducible on any machine and independent of any
C++ ``<random>`` implementation.

* ``uniform_cloud``  -- cloud "U": uniform in ``[0, scale)^dim``.
* ``lidar_cloud``    -- cloud "L": a terrestrial scanner inside an 80x80x15 room
  (rays with uniform azimuth, elevation in [-60deg, +40deg], first hit on
  floor / ceiling / walls, range cap 60, Gaussian range noise sigma = 0.01),
  multiplied by the unit scale S (default 20).
* ``morton_order``   -- permutation sorting 3-D points along a 30-bit Z curve
  (the "coherent" query order of BASELINE.md section 2).
"""

from __future__ import annotations

import os

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def raw_uniform24(seed: int, count: int, stream: int = 0) -> np.ndarray:
    """``count`` floats in [0, 1) with 24 random mantissa bits, float32-exact."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed * 0x10001 + stream * 0x9E37 + 1], dtype=np.uint64))[0]
        ctr = np.arange(count, dtype=np.uint64) + base
    bits = _splitmix64(ctr) >> np.uint64(40)
    return (bits.astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def uniform_cloud(n: int, dim: int = 3, seed: int = 1, scale: float = 1.0) -> np.ndarray:
    """Cloud U: ``n`` points uniform in ``[0, scale)^dim`` (float32, C order)."""
    u = raw_uniform24(seed, n * dim).reshape(n, dim)
    return np.ascontiguousarray(u * np.float32(scale), dtype=np.float32)


def _normal(seed: int, count: int, stream: int) -> np.ndarray:
    """Box-Muller on two raw uniform streams (float64 math, deterministic)."""
    u1 = raw_uniform24(seed, count, stream).astype(np.float64)
    u2 = raw_uniform24(seed, count, stream + 1).astype(np.float64)
    u1 = np.maximum(u1, 2.0 ** -25)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def lidar_cloud(n: int, seed: int = 1, pose=(0.0, 0.0), unit_scale: float = 20.0) -> np.ndarray:
    """Cloud L: LiDAR-like room scan from scanner position ``pose`` (x, y)."""
    room_min = np.array([-40.0, -40.0, 0.0])
    room_max = np.array([40.0, 40.0, 15.0])
    origin = np.array([pose[0], pose[1], 2.0])
    az = raw_uniform24(seed, n, 10).astype(np.float64) * (2.0 * np.pi)
    el_lo, el_hi = np.deg2rad(-60.0), np.deg2rad(40.0)
    el = el_lo + raw_uniform24(seed, n, 11).astype(np.float64) * (el_hi - el_lo)
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_lo = (room_min - origin) / d
        t_hi = (room_max - origin) / d
    t_exit = np.where(d > 0, t_hi, t_lo)
    t_exit = np.where(d == 0, np.inf, t_exit)
    t = np.min(t_exit, axis=1)
    t = np.minimum(t, 60.0)
    t = t + 0.01 * _normal(seed, n, 12)
    pts = origin + d * t[:, None]
    return np.ascontiguousarray(pts * unit_scale, dtype=np.float32)


def _part1by2(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64) & np.uint64(0x3FF)
    x = (x | (x << np.uint64(16))) & np.uint64(0x030000FF)
    x = (x | (x << np.uint64(8))) & np.uint64(0x0300F00F)
    x = (x | (x << np.uint64(4))) & np.uint64(0x030C30C3)
    x = (x | (x << np.uint64(2))) & np.uint64(0x09249249)
    return x


def morton_order(points: np.ndarray) -> np.ndarray:
    """Stable permutation sorting ``points`` (n, <=3) by a 30-bit Morton code."""
    p = np.asarray(points, dtype=np.float64)
    lo = p.min(axis=0)
    ext = np.maximum(p.max(axis=0) - lo, 1e-30)
    cells = np.minimum(((p - lo) / ext * 1024.0).astype(np.int64), 1023)
    code = np.zeros(len(p), dtype=np.uint64)
    for d in range(min(p.shape[1], 3)):
        code |= _part1by2(cells[:, d]) << np.uint64(d)
    return np.argsort(code, kind="stable")


#: BASELINE.json configs[1]: the README's tree / query sizes.
CONFIG2_N = 7_733_372
CONFIG2_NQ = 7_200_863


def config2_clouds(cloud: str = "L", n: int = CONFIG2_N, nq: int = CONFIG2_NQ):
    """(tree points, query points) for BASELINE config 2, cloud "U" or "L"."""
    if cloud == "U":
        return uniform_cloud(n, 3, seed=1, scale=100.0), uniform_cloud(nq, 3, seed=2, scale=100.0)
    if cloud == "L":
        return (lidar_cloud(n, seed=1, pose=(0.0, 0.0), unit_scale=20.0),
                lidar_cloud(nq, seed=2, pose=(3.0, 1.5), unit_scale=20.0))
    raise ValueError("cloud must be 'U' or 'L'")


def sift_like_cloud(n: int, dim: int = 128, seed: int = 1, centres: int = 1000, centre_seed: int = 77,
                    sigma: float = 30.0) -> np.ndarray:
    """Synthetic stand-in for SIFT descriptors (BASELINE config 5; SIFT-1M cannot be downloaded
    here): a mixture of ``centres`` Gaussians (centres uniform in [40, 180]^dim, sigma 30), clipped
    to [0, 218] and rounded to integers like SIFT bytes.  ``centre_seed`` is shared by the point
    and query clouds so they come from the same mixture."""
    crng = np.random.Generator(np.random.PCG64(centre_seed))
    c = crng.uniform(40.0, 180.0, size=(centres, dim)).astype(np.float32)
    rng = np.random.Generator(np.random.PCG64(seed))
    which = rng.integers(0, centres, size=n)
    x = c[which] + rng.normal(0.0, sigma, size=(n, dim)).astype(np.float32)
    return np.ascontiguousarray(np.rint(np.clip(x, 0.0, 218.0)).astype(np.float32))


# ---- real data: the reference's own input files ---------------------------------------------------
# (None of these ship here -- there is no network -- but a user who has the Bremen scans converted
# by the reference's uosr_to_bin tool, or SIFT-1M from corpus-texmex, can hand them to bench.py
# --points / --queries or to tools/bench_forest.py.)

def read_bin(path: str, dim: int = 3, dtype=np.float32) -> np.ndarray:
    """A file written by the reference's ``write_bin`` (pico_toolshed/format/format_bin.hpp:9-32):
    the raw elements of a ``std::vector<point>`` back to back, no header -- for its benchmark
    (examples/benchmark/benchmark.hpp:28-31: ``scans0.bin`` / ``scans1.bin``) float32 triples.
    Returns ``(n, dim)``; trailing bytes that do not make a whole point are ignored, as
    ``file_size / sizeof(T)`` does there."""
    dt = np.dtype(dtype)
    n = os.path.getsize(path) // (dt.itemsize * dim)
    return np.ascontiguousarray(np.fromfile(path, dtype=dt, count=n * dim).reshape(n, dim))


def write_bin(path: str, points: np.ndarray) -> None:
    """The inverse of :func:`read_bin` (format_bin.hpp:9-19)."""
    np.ascontiguousarray(points).tofile(path)


def read_xvecs(path: str) -> np.ndarray:
    """``.fvecs`` / ``.bvecs`` / ``.ivecs`` of corpus-texmex as the reference reads them
    (pico_toolshed/format/format_xvecs.hpp:41-67): every row is an int32 component count followed
    by that many float32 / uint8 / int32 values; the element type comes from the file name.
    Returns ``(rows, dim)`` in the file's element type (convert ``bvecs`` with ``astype(float32)``
    for a tree)."""
    low = path.lower()
    if low.endswith(".fvecs"):
        dt = np.dtype("<f4")
    elif low.endswith(".bvecs"):
        dt = np.dtype("u1")
    elif low.endswith(".ivecs"):
        dt = np.dtype("<i4")
    else:
        raise ValueError("filename expected to end with .fvecs, .bvecs or .ivecs")
    size = os.path.getsize(path)
    if size < 4:
        return np.empty((0, 0), dtype=dt)
    dim = int(np.fromfile(path, dtype="<i4", count=1)[0])
    if dim <= 0:
        raise ValueError("bad component count in the first row")
    row = 4 + dt.itemsize * dim
    rows = size // row
    raw = np.fromfile(path, dtype=np.uint8, count=rows * row).reshape(rows, row)
    if not np.all(raw[:, :4].copy().view("<i4")[:, 0] == dim):
        raise ValueError("rows of different lengths")
    return np.ascontiguousarray(raw[:, 4:]).view(dt).reshape(rows, dim)


def write_xvecs(path: str, rows: np.ndarray) -> None:
    """Writes ``rows`` in the format :func:`read_xvecs` reads (element type from the file name)."""
    low = path.lower()
    dt = np.dtype("<f4") if low.endswith(".fvecs") else np.dtype("u1") if low.endswith(".bvecs") else np.dtype("<i4")
    r = np.ascontiguousarray(rows, dtype=dt)
    out = np.empty((r.shape[0], 4 + dt.itemsize * r.shape[1]), dtype=np.uint8)
    out[:, :4] = np.full(r.shape[0], r.shape[1], dtype="<i4").view(np.uint8).reshape(-1, 4)
    out[:, 4:] = r.view(np.uint8).reshape(r.shape[0], -1)
    out.tofile(path)


def load_points(path: str, dim: int = 3) -> np.ndarray:
    """float32 ``(n, dim)`` points from a ``.bin`` (reference ``write_bin``), ``.fvecs`` / ``.bvecs`` or
    ``.npy`` file."""
    low = path.lower()
    if low.endswith((".fvecs", ".bvecs", ".ivecs")):
        return np.ascontiguousarray(read_xvecs(path), dtype=np.float32)
    if low.endswith(".npy"):
        return np.ascontiguousarray(np.load(path), dtype=np.float32)
    return read_bin(path, dim)
