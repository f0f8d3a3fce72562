"""Builds libptk.so (the HIP backend + C ABI) in-tree for gfx950.

``python -m pico_tree_amd.build`` or ``pico_tree_amd.build.build()``.  hipcc
cross-compiles without a GPU, so this also runs in CPU-only containers.  The
shared object lands next to the sources (``pico_tree_amd/csrc/libptk.so``) so it
travels with the repository snapshot to the GPU box.
"""

from __future__ import annotations

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libptk.so")
SOURCES = [os.path.join(CSRC, "ptk_backend.hip")]
def _headers():
    """Everything the translation unit may include: the kernels and host helpers next to it, the C ABI and the
    header-only host API (the builder, the stream format and the per-query searches are compiled into the library)."""
    found = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")]
    for base, _, files in os.walk(os.path.join(ROOT, "include")):
        found += [os.path.join(base, f) for f in sorted(files) if f.endswith((".h", ".hpp"))]
    return found


HEADERS = _headers()

#: -ffp-contract=off: the results contract forbids fused multiply-add
#: (contraction changes distance bits).
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fPIC",
    "-shared",
    "-Wall",
    "-Wno-unused-result",
]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > built for p in SOURCES + HEADERS if os.path.exists(p))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile if needed; returns the path of libptk.so."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + os.environ.get("PTK_EXTRA_FLAGS", "").split() + [  # -D... of kernel experiments
        "-I" + os.path.join(ROOT, "include"),
        "-I" + CSRC,
        "-o", LIB,
    ] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
