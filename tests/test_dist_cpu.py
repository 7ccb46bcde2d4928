"""CPU, world_size 2 over gloo: the multi-process plumbing of the batched mode — stream sharding, the map broadcast
(the path's only collective) and the max-over-ranks timing rule."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from loam_velodyne_amd import dist as lxdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = lxdist.stream_ids(rank, world, 3)
    m = torch.zeros((1000, 4), dtype=torch.float32)
    if rank == 0:
        m.copy_(torch.arange(4000, dtype=torch.float32).reshape(1000, 4))
    lxdist.broadcast_map(m, dist)
    t = lxdist.max_over_ranks(1.0 + rank, dist)
    poses = np.full((3, 6), float(rank), np.float32)
    allp = lxdist.gather_poses(poses, dist)
    q.put((rank, ids, float(m.sum()), t, allp.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_broadcast_and_timing_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2] and res[1][1] == [3, 4, 5]          # disjoint contiguous shards
    assert res[0][2] == res[1][2] == float(sum(range(4000)))          # every rank holds the same map
    assert res[0][3] == res[1][3] == 2.0                              # job time = slowest rank
    assert np.array(res[0][4]).shape == (6, 6) and np.array(res[1][4])[3:, 0].tolist() == [1.0, 1.0, 1.0]


def _gather_worker(rank, world, port, batch, q):
    """the library's own shard rule and record layout (loamx_dist_shard_of / pack_results / unpack_results) over a gloo all-gather:
    what loamx_dist_allgather_results does with RCCL, transport swapped"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from loam_velodyne_amd import loamx
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b0, b1 = loamx.dist_shard_of(rank, world, batch)
    ids = np.arange(b0, b1)
    poses = (ids[:, None] * 10 + np.arange(6)[None, :]).astype(np.float32)       # record i carries its global index
    flags = np.stack([ids + 100, ids + 200], 1).astype(np.int32)
    cnt = torch.tensor([b1 - b0], dtype=torch.int64)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, cnt)                                                    # 1. the counts
    counts = np.array([int(c.item()) for c in counts], np.uint32)
    n_pad = int(counts.max())
    out = None
    if n_pad:                                                                       # (all ranks see the same counts)
        send = torch.from_numpy(loamx.dist_pack_results(poses, flags, n_pad))       # 2. the padded records
        recv = [torch.zeros_like(send) for _ in range(world)]
        dist.all_gather(recv, send)
        out = loamx.dist_unpack_results(torch.stack(recv).numpy(), counts, n_pad)
    q.put((rank, [b0, b1], counts.tolist(), None if out is None else out[0].tolist(), None if out is None else out[1].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [5, 1, 4, 0])
def test_unequal_shards_gather_world2(batch):
    """B not divisible by G (and B < G: an empty shard that still takes part; B = 0: nobody sends)"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1][0] == 0 and res[0][1][1] == res[1][1][0] and res[1][1][1] == batch          # contiguous, covering
    want_p = (np.arange(batch)[:, None] * 10 + np.arange(6)[None, :]).astype(np.float32)
    want_f = np.stack([np.arange(batch) + 100, np.arange(batch) + 200], 1)
    for r in res:
        assert r[2] == [res[0][1][1] - res[0][1][0], res[1][1][1] - res[1][1][0]]
        if batch:
            assert np.array_equal(np.array(r[3], np.float32), want_p) and np.array_equal(np.array(r[4]), want_f)   # batch order, on every rank
        else:
            assert r[3] is None


def test_shard_rule_and_layout_many_ranks():
    """the same functions for world sizes that cannot run here: (10, 4) -> 2/3/2/3, (3, 8) -> five empty shards"""
    from loam_velodyne_amd import loamx
    for batch, world in ((10, 4), (3, 8), (32, 4), (64, 8), (7, 7)):
        shards = [loamx.dist_shard_of(r, world, batch) for r in range(world)]
        assert shards[0][0] == 0 and shards[-1][1] == batch and all(shards[r][1] == shards[r + 1][0] for r in range(world - 1))
        counts = np.array([e - b for b, e in shards], np.uint32)
        assert counts.max() - counts.min() <= 1
        n_pad = int(counts.max())
        blocks = []
        for b, e in shards:
            ids = np.arange(b, e)
            blocks.append(loamx.dist_pack_results((ids[:, None] + np.arange(6)[None, :] * 0.5).astype(np.float32), np.stack([ids, -ids], 1).astype(np.int32), n_pad))
        poses, flags = loamx.dist_unpack_results(np.stack(blocks), counts, n_pad)
        ids = np.arange(batch)
        assert np.array_equal(poses, (ids[:, None] + np.arange(6)[None, :] * 0.5).astype(np.float32)) and np.array_equal(flags, np.stack([ids, -ids], 1))
    with pytest.raises(loamx.LoamxError):
        loamx.dist_pack_results(np.zeros((3, 6), np.float32), None, 2)            # more records than the padded size


def test_stream_layout_helpers():
    from loam_velodyne_amd import dist as lxdist
    all_ids = sum((lxdist.stream_ids(r, 4, 8) for r in range(4)), [])
    assert all_ids == list(range(32))                                  # BASELINE configs[3]: batch 32 over 4 GPUs
    starts = {lxdist.stream_start(g) for g in range(64)}
    assert len(starts) == 64
    assert lxdist.split_map(1_000_000) == (100_000, 900_000)
