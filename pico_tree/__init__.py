"""``import pico_tree as pt`` -- the module name of the reference's Python binding
(/root/reference/src/pyco_tree/pico_tree/__init__.py exports DArray, Metric, KdTree, load_kd_tree,
save_kd_tree), served by :mod:`pico_tree_amd`: a script written for the reference, e.g. its
``examples/python/kd_tree.py``, runs unchanged, with its batched searches on the MI355X.

Everything else :mod:`pico_tree_amd` offers (``KdForest``, ``MultiKdTree``, device tensors, ...) is
reachable under this name as well.
"""

from pico_tree_amd import *  # noqa: F401,F403
from pico_tree_amd import (DArray, KdTree, Metric, load_kd_tree, save_kd_tree,  # noqa: F401
                           __all__ as _amd_all)

__all__ = ["DArray", "Metric", "KdTree", "load_kd_tree", "save_kd_tree"] + [n for n in _amd_all if n not in (
    "DArray", "Metric", "KdTree", "load_kd_tree", "save_kd_tree")]
