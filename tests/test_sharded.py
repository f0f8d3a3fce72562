"""Multi-process path on CPU: two ranks over gloo run the shard / search / gather
bookkeeping of pico_tree_amd.sharded (the same code bench.py drives over RCCL), with the
oracle standing in for the per-shard GPU search, and rank 0 must end up with exactly the
unsharded answer in caller order."""

from __future__ import annotations

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pico_tree_amd.sharded import padded_shard, shard_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_every_row_once():
    for nq in (0, 1, 7, 64, 1000, 7_200_863):
        for world in (1, 2, 3, 8):
            shards = [shard_of(nq, world, r) for r in range(world)]
            assert shards[0].lo == 0 and shards[-1].hi == nq
            for a, b in zip(shards, shards[1:]):
                assert a.hi == b.lo
            assert all(s.rows <= s.per for s in shards)
            assert sum(s.rows for s in shards) == nq
    sh = shard_of(10, 4, 3)
    q = np.arange(30, dtype=np.float32).reshape(10, 3)
    p = padded_shard(q, sh)
    assert p.shape == (3, 3) and np.array_equal(p[0], q[9]) and np.array_equal(p[2], q[9])


def _worker(rank, world, port, nq, k, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from pico_tree_amd import datasets as ds
    from pico_tree_amd.sharded import ShardedSearch, padded_shard, shard_of

    pts = ds.uniform_cloud(20_000, 3, seed=5)   # every rank builds the same replica
    ref = oracle.Oracle(pts, 10, "port")

    def search(q_local, out):  # stand-in for tree.search_knn(q_local, k, out).raw on the GPU
        res = ref.search_knn(q_local.numpy(), k)
        out.copy_(torch.from_numpy(res.view(np.int32).reshape(len(res), k, 2).copy()))
        return out

    sh = shard_of(nq, world, rank)
    ss = ShardedSearch(sh, search, lambda: torch.empty((sh.per, k, 2), dtype=torch.int32), depth=2)
    ok = True
    # Several steps with DIFFERENT batches: the gather of one step overlaps the next search, and the
    # rotating buffers must never be overwritten while a gather still reads them.
    for step in range(5):
        q = ds.uniform_cloud(nq, 3, seed=6 + step)
        ss.step(torch.from_numpy(padded_shard(q, sh)))
        if step in (0, 3, 4):
            full = ss.result()
            if rank == 0:
                want = ref.search_knn(q, k).view(np.int32).reshape(nq, k, 2)
                ok = ok and bool(np.array_equal(full.numpy(), want))
            else:
                assert full is None
    # Replicated batches (weak scaling): every rank searches its own nq rows, rank 0 gets world * nq.
    sh1 = shard_of(nq * world, world, rank)
    ss2 = ShardedSearch(sh1, search, lambda: torch.empty((nq, k, 2), dtype=torch.int32))
    mine = ds.uniform_cloud(nq, 3, seed=100 + rank)
    ss2.step(torch.from_numpy(mine))
    both = ss2.result(rows_per_rank=nq)
    if rank == 0:
        want = np.concatenate([ref.search_knn(ds.uniform_cloud(nq, 3, seed=100 + r), k) for r in range(world)])
        ok = ok and bool(np.array_equal(both.numpy(), want.view(np.int32).reshape(nq * world, k, 2)))
        np.save(os.path.join(tmp, "ok.npy"), np.array([int(ok)]))
    # Ragged results: radius rows gathered in two steps (counts, then one transfer per rank).
    from pico_tree_amd.sharded import sharded_radius
    radius = 0.0012
    for nq_r in (nq, 3):  # 3 rows over 2 ranks: the last shard is short, and may be empty of hits
        q = ds.uniform_cloud(nq_r, 3, seed=50)
        shr = shard_of(nq_r, world, rank)

        def search_radius(q_local):  # stand-in for tree.search_radius_device on the GPU
            off, flat = ref.search_radius(q_local.numpy(), radius)
            return (torch.from_numpy(off.astype(np.int64)),
                    torch.from_numpy(flat.view(np.int32).reshape(-1, 2).copy()))

        res = sharded_radius(shr, search_radius, torch.from_numpy(padded_shard(q, shr)))
        if rank == 0:
            want_off, want = ref.search_radius(q, radius)
            ok = ok and bool(np.array_equal(res[0].numpy().astype(np.uint64), want_off))
            ok = ok and res[1].numpy().tobytes() == want.tobytes()
            np.save(os.path.join(tmp, "ok.npy"), np.array([int(ok)]))
        else:
            assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq,k", [(1001, 1), (4096, 3)])
def test_two_ranks_gather_matches_unsharded(tmp_path, nq, k):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, nq, k, str(tmp_path)), nprocs=2, join=True)
    assert np.load(os.path.join(tmp_path, "ok.npy"))[0] == 1
