"""Query-batch sharding across the GPUs of one node (one process per GPU).

The path shards naturally: queries are independent (the reference's only batched
caller is an OpenMP loop over rows, ``_pyco_tree/kd_tree.hpp:128-134``).  Every rank
keeps a full replica of the tree (~143 MB for BASELINE config 2, against 288 GB of
HBM), takes one contiguous range of ``ceil(nq / world)`` query rows, searches it, and
rank 0 collects the ``(index, distance)`` records with ONE gather over RCCL/xGMI
(each peer reaches the root over its own point-to-point link; 7.2 MB per rank at
k = 1, so no ring or tree is needed).  Rows keep the caller's order: shard ``r``
owns rows ``[r * per, min((r + 1) * per, nq))``.

Nothing here touches the search itself -- ``search`` is any callable filling a
``(per, k, 2)`` int32 tensor from a ``(per, dim)`` float32 tensor (on the GPU:
``lambda q, out: tree.search_knn(q, k, out).raw``), which is what lets the 2-process gloo
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
    """Search the local rows, then gather every rank's rows on rank 0.

    ``search(q, out)`` must fill (and return) the preallocated ``out``.  The gather of step i is
    issued asynchronously and overlaps the search of step i + 1: ``depth`` output buffers are used
    in rotation and a buffer is only handed to a new search after the gather that reads it has been
    waited for (``work.wait()`` orders the current stream behind the collective; on RCCL it does
    not block the host).  ``finish()`` waits for everything still in flight -- call it before
    stopping a clock.
    """

    def __init__(self, shard: Shard, search, make_out, group=None, depth: int = 2):
        self.shard = shard
        self.search = search
        self.group = group
        self.depth = max(1, int(depth))
        self._out = [make_out() for _ in range(self.depth)]
        self._recv = [None] * self.depth     # rank 0: list of world tensors per buffer
        self._work = [None] * self.depth
        self._step = 0
        self._last = None

    def step(self, q_local):
        """One pass: returns the local result tensor of this step."""
        import torch.distributed as dist

        b = self._step % self.depth
        self._step += 1
        if self._work[b] is not None:  # the gather that still reads this buffer
            self._work[b].wait()
            self._work[b] = None
        res = self.search(q_local, self._out[b])
        if self.shard.world > 1:
            if self.shard.rank == 0:
                if self._recv[b] is None:
                    import torch
                    self._recv[b] = [torch.empty_like(res) for _ in range(self.shard.world)]
                self._work[b] = dist.gather(res, self._recv[b], dst=0, group=self.group, async_op=True)
            else:
                self._work[b] = dist.gather(res, None, dst=0, group=self.group, async_op=True)
        self._last = b
        return res

    def finish(self) -> None:
        for b in range(self.depth):
            if self._work[b] is not None:
                self._work[b].wait()
                self._work[b] = None

    def result(self, rows_per_rank=None):
        """Rank 0: the rows of the LAST step from every rank, concatenated in rank order and cut to
        ``shard.nq`` rows (contiguous sharding) or to ``rows_per_rank`` rows each (replicated
        batches); other ranks: ``None``."""
        import torch

        self.finish()
        b = self._last
        if self.shard.world == 1:
            return self._out[b][:self.shard.nq]
        if self.shard.rank != 0:
            return None
        if rows_per_rank is not None:
            return torch.cat([t[:rows_per_rank] for t in self._recv[b]], dim=0)
        return torch.cat(self._recv[b], dim=0)[:self.shard.nq]


def sharded_radius(shard: Shard, search, q_local, group=None):
    """Radius search of a sharded batch: ragged rows, so the gather is two-step.

    ``search(q_local)`` returns this rank's ``(offsets, rows)`` -- ``offsets`` an int64 tensor of
    ``rows_local + 1`` entries, ``rows`` an int32 ``(total, 2)`` tensor of (index, distance bits) in
    the reference's traversal order (on the GPU: ``tree.search_radius_device``).  ``q_local`` holds
    ``shard.per`` rows (``padded_shard``); the padding rows are cut before anything is sent.

    Step 1 gathers the per-row counts (fixed size: ``shard.per`` per rank) on rank 0, which gives it
    every rank's payload size; step 2 moves each payload with one point-to-point transfer into its
    place of the final buffer (shards are contiguous row ranges, so rank order is row order).
    Returns ``(offsets[nq + 1], rows)`` on rank 0 and ``None`` elsewhere."""
    import torch
    import torch.distributed as dist

    offsets, rows = search(q_local)
    n_local = shard.rows
    counts = (offsets[1:n_local + 1] - offsets[:n_local]).to(torch.int64)
    rows = rows[:int(offsets[n_local].item())].contiguous()
    if shard.world == 1:
        out_off = torch.zeros(shard.nq + 1, dtype=torch.int64, device=counts.device)
        out_off[1:] = torch.cumsum(counts, 0)
        return out_off, rows
    padded = torch.zeros(shard.per, dtype=torch.int64, device=counts.device)
    padded[:n_local] = counts
    if shard.rank == 0:
        parts = [torch.empty_like(padded) for _ in range(shard.world)]
        dist.gather(padded, parts, dst=0, group=group)
        all_counts = torch.cat(parts)[:shard.nq]
        out_off = torch.zeros(shard.nq + 1, dtype=torch.int64, device=counts.device)
        out_off[1:] = torch.cumsum(all_counts, 0)
        total = int(out_off[-1].item())
        out = torch.empty((total, 2), dtype=torch.int32, device=rows.device)
        out[:rows.shape[0]] = rows
        for r in range(1, shard.world):
            lo = min(r * shard.per, shard.nq)
            hi = min(lo + shard.per, shard.nq)
            a, b = int(out_off[lo].item()), int(out_off[hi].item())
            if b > a:
                dist.recv(out[a:b], src=r, group=group)
        return out_off, out
    dist.gather(padded, None, dst=0, group=group)
    if rows.shape[0] > 0:
        dist.send(rows, dst=0, group=group)
    return None
