"""Builds libptk.so (the HIP backend + C ABI) in-tree for gfx950.

``python -m pico_tree_amd.build`` or ``pico_tree_amd.build.build()``.  hipcc
cross-compiles without a GPU, so this also runs in CPU-only containers.  The
shared object lands next to the sources (``pico_tree_amd/csrc/libptk.so``) so it
travels with the repository snapshot to the GPU box.

The library is one translation unit per kernel family (``csrc/ptk_backend_core.hpp``
lists them): the units compile side by side, each into ``csrc/_obj/<unit>.o`` with a
compiler-written dependency file next to it, and only the units that include what
changed are compiled again (an edit of ``ptk_kernels_lists.hpp`` rebuilds the radius
unit and the core, not the k-NN, any-dimension, topological or double units).
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


def _unit_stale(unit: str, flags_line: str) -> bool:
    obj = os.path.join(OBJ, unit + ".o")
    dep = os.path.join(OBJ, unit + ".d")
    cmdfile = os.path.join(OBJ, unit + ".cmd")
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(cmdfile)):
        return True
    with open(cmdfile) as f:
        if f.read() != flags_line:  # other flags (an experiment build): compile again
            return True
    built = os.path.getmtime(obj)
    deps = _deps(dep)
    if not deps:
        return True
    for p in deps:
        try:
            if os.path.getmtime(p) > built:
                return True
        except OSError:  # a header that is gone
            return True
    return False


def _compile(unit: str, flags: list, flags_line: str, verbose: bool) -> None:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = os.path.join(OBJ, unit + ".o")
    cmd = [hipcc] + flags + ["-c", "-MD", "-MF", os.path.join(OBJ, unit + ".d"), "-o", obj,
                             os.path.join(CSRC, unit + ".hip")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(os.path.join(OBJ, unit + ".cmd"), "w") as f:
        f.write(flags_line)


def build(force: bool = False, verbose: bool = False, lib: str = LIB) -> str:
    """Compile what is out of date and link; returns the path of libptk.so."""
    os.makedirs(OBJ, exist_ok=True)
    flags = _compile_flags()
    flags_line = " ".join(flags)
    todo = [u for u in UNITS if force or _unit_stale(u, flags_line)]
    if todo:
        jobs = max(1, min(len(todo), int(os.environ.get("PTK_BUILD_JOBS", os.cpu_count() or 4))))
        with ThreadPoolExecutor(max_workers=jobs) as pool:
            for _ in pool.map(lambda u: _compile(u, flags, flags_line, verbose), todo):
                pass
    objs = [os.path.join(OBJ, u + ".o") for u in UNITS]
    if todo or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
