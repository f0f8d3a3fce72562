"""Query-batch sharding across the GPUs of one node (one process per GPU).

The path shards naturally: queries are independent (the reference's only batched
caller is an OpenMP loop over rows, ``_pyco_tree/kd_tree.hpp:128-134``).  Every rank
keeps a full replica of the tree (~143 MB for BASELINE config 2, against 288 GB of
HBM), takes one contiguous range of ``ceil(nq / world)`` query rows, searches it, and
rank 0 collects the ``(index, distance)`` records with ONE gather over RCCL/xGMI
(each peer reaches the root over its own point-to-point link; 7.2 MB per rank at
k = 1, so no ring or tree is needed).  Rows keep the caller's order: shard ``r``
owns rows ``[r * per, min((r + 1) * per, nq))``.

Nothing here touches the search itself -- ``search`` is any callable mapping a
``(per, dim)`` float32 tensor to a ``(per, k, 2)`` int32 tensor (on the GPU:
``lambda q: tree.search_knn(q, k).raw``), which is what lets the 2-process gloo
test exercise the bookkeeping on CPU.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    nq: int
    per: int   # rows per rank (the last shard is padded up to this)
    lo: int    # first row owned by this rank
    hi: int    # one past the last row owned by this rank

    @property
    def rows(self) -> int:
        return self.hi - self.lo


def shard_of(nq: int, world: int, rank: int) -> Shard:
    """Contiguous split of ``nq`` rows over ``world`` ranks."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world / rank")
    per = (nq + world - 1) // world if nq else 0
    lo = min(rank * per, nq)
    hi = min(lo + per, nq)
    return Shard(rank, world, nq, per, lo, hi)


def padded_shard(queries: np.ndarray, shard: Shard) -> np.ndarray:
    """This rank's rows, padded to ``shard.per`` rows by repeating a valid row, so that
    every rank contributes an equally sized block to the gather."""
    dim = queries.shape[1]
    out = np.empty((shard.per, dim), dtype=np.float32)
    out[:shard.rows] = queries[shard.lo:shard.hi]
    if shard.rows < shard.per:
        out[shard.rows:] = queries[shard.lo] if shard.rows else queries[0]
    return out


class ShardedSearch:
    """Search the local shard, then gather every shard's rows on rank 0."""

    def __init__(self, shard: Shard, search, group=None):
        self.shard = shard
        self.search = search
        self.group = group
        self._gathered = None

    def step(self, q_local, out=None):
        """One pass: returns the local result tensor; rank 0 also refreshes ``gathered``."""
        import torch.distributed as dist

        res = self.search(q_local) if out is None else self.search(q_local, out)
        if self.shard.world > 1:
            if self.shard.rank == 0:
                if self._gathered is None:
                    import torch
                    self._gathered = [torch.empty_like(res) for _ in range(self.shard.world)]
                dist.gather(res, self._gathered, dst=0, group=self.group)
            else:
                dist.gather(res, None, dst=0, group=self.group)
        return res

    def result(self, local):
        """Rank 0: all ``nq`` rows in caller order, ``(nq, k, 2)``; other ranks: ``None``."""
        import torch

        if self.shard.world == 1:
            return local[:self.shard.nq]
        if self.shard.rank != 0:
            return None
        return torch.cat(self._gathered, dim=0)[:self.shard.nq]
