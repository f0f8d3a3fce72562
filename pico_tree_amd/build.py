"""Builds libptk.so (the HIP backend + C ABI) in-tree for gfx950.

``python -m pico_tree_amd.build`` or ``pico_tree_amd.build.build()``.  hipcc
cross-compiles without a GPU, so this also runs in CPU-only containers.  The
shared object lands next to the sources (``pico_tree_amd/csrc/libptk.so``) so it
travels with the repository snapshot to the GPU box.

The library is one translation unit per kernel family (``csrc/ptk_backend_core.hpp``
lists them): the units compile side by side, each into ``csrc/_obj/<unit>.o`` with a
compiler-written dependency file next to it, and only the units that include what
changed are compiled again (an edit of ``ptk_kernels_lists.hpp`` rebuilds the radius
unit and the core, not the k-NN, any-dimension, topological or double units).  "Changed"
is decided by CONTENT (a SHA-256 of the flags and of every repository file a unit
includes, kept next to its object), not by modification times: the objects travel with
the snapshot to the GPU box, where the copy has shuffled the times, and nothing is
compiled again there.
"""

from __future__ import annotations

import os
import shlex
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libptk.so")
OBJ = os.path.join(CSRC, "_obj")
#: the translation units of the library, the longest to compile first
UNITS = ["ptk_family_knn", "ptk_backend", "ptk_family_nd", "ptk_family_radius", "ptk_family_f64", "ptk_family_topo"]
SOURCES = [os.path.join(CSRC, u + ".hip") for u in UNITS]

#: -ffp-contract=off: the results contract forbids fused multiply-add
#: (contraction changes distance bits).
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fPIC",
    "-Wall",
    "-Wno-unused-result",
    "-Wno-unused-function",  # (a unit does not use every inline helper or static kernel of the shared headers)
]


def _compile_flags() -> list:
    return FLAGS + os.environ.get("PTK_EXTRA_FLAGS", "").split() + [  # -D... of kernel experiments
        "-I" + os.path.join(ROOT, "include"),
        "-I" + CSRC,
    ]


def _deps(depfile: str) -> list:
    """The prerequisites a compiler-written dependency file (-MD) lists."""
    try:
        with open(depfile) as f:
            text = f.read()
    except OSError:
        return []
    text = text.replace("\\\n", " ")
    _, _, rest = text.partition(":")
    return shlex.split(rest)


def _signature(unit: str, flags_line: str) -> str:
    """What an object was made from: the flags and the CONTENTS of every repository file its dependency file lists (the
    toolchain's own headers are taken as given).  Contents, not modification times: a copy of the tree -- the snapshot
    that travels to the GPU box -- keeps the bytes and shuffles the times, and a unit must not be compiled again there."""
    import hashlib

    # (the flags without the place the tree happens to lie at: the GPU box runs a copy under another path)
    h = hashlib.sha256(flags_line.replace(os.path.realpath(ROOT), "<root>").replace(ROOT, "<root>").encode())
    deps = sorted(set(os.path.realpath(p) for p in _deps(os.path.join(OBJ, unit + ".d"))))
    root = os.path.realpath(ROOT) + os.sep
    mine = [p for p in deps if p.startswith(root)]
    if not mine:
        return ""
    for p in mine:
        try:
            with open(p, "rb") as f:
                h.update(p[len(root):].encode() + b"\0" + f.read())
        except OSError:  # a header that is gone
            return ""
    return h.hexdigest()


def _unit_stale(unit: str, flags_line: str) -> bool:
    obj = os.path.join(OBJ, unit + ".o")
    sigfile = os.path.join(OBJ, unit + ".sig")
    if not (os.path.exists(obj) and os.path.exists(sigfile)):
        return True
    with open(sigfile) as f:
        recorded = f.read().strip()
    return recorded == "" or recorded != _signature(unit, flags_line)


def _compile(unit: str, flags: list, flags_line: str, verbose: bool) -> None:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = os.path.join(OBJ, unit + ".o")
    cmd = [hipcc] + flags + ["-c", "-MD", "-MF", os.path.join(OBJ, unit + ".d"), "-o", obj,
                             os.path.join(CSRC, unit + ".hip")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(os.path.join(OBJ, unit + ".sig"), "w") as f:
        f.write(_signature(unit, flags_line))


def build(force: bool = False, verbose: bool = False, lib: str = LIB) -> str:
    """Compile what is out of date and link; returns the path of libptk.so."""
    flags = _compile_flags()
    flags_line = " ".join(flags)
    os.makedirs(OBJ, exist_ok=True)
    todo = [u for u in UNITS if force or _unit_stale(u, flags_line)]
    if todo:
        jobs = max(1, min(len(todo), int(os.environ.get("PTK_BUILD_JOBS", os.cpu_count() or 4))))
        with ThreadPoolExecutor(max_workers=jobs) as pool:
            for _ in pool.map(lambda u: _compile(u, flags, flags_line, verbose), todo):
                pass
    objs = [os.path.join(OBJ, u + ".o") for u in UNITS]
    # (the link is remembered the same way: the signatures of the objects the library was linked from)
    linked = os.path.join(OBJ, os.path.basename(lib) + ".sig")
    want = "\n".join(open(os.path.join(OBJ, u + ".sig")).read() for u in UNITS)
    have = open(linked).read() if os.path.exists(linked) else None
    if todo or not os.path.exists(lib) or have != want:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(linked, "w") as f:
            f.write(want)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
