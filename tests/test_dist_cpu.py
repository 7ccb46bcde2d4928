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


def _cloud_worker(rank, world, port, q):
    """the epoch's merge step over gloo: every rank packs its streams' clouds + poses with the library's own layout rule
    (loamx_dist_pack_clouds), the words travel to rank 0 counts first (what loamx_dist_gatherv does with ncclSend / ncclRecv), rank 0
    splits the buffer by the counts and unpacks (loamx_dist_unpack_clouds_*)"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from loam_velodyne_amd import loamx
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n_streams = [3, 0, 2][rank % 3]                       # rank 1 holds NO stream: an empty message that still takes part
    corners = [rng.normal(size=(int(rng.integers(0, 40)), 4)).astype(np.float32) for _ in range(n_streams)]
    surfs = [rng.normal(size=(int(rng.integers(1, 90)), 4)).astype(np.float32) for _ in range(n_streams)]
    if n_streams:
        corners[0] = np.zeros((0, 4), np.float32)        # an empty cloud inside a message
    poses = rng.normal(size=(n_streams, 6)).astype(np.float32)
    words = loamx.dist_pack_clouds(corners, surfs, poses) if n_streams else np.zeros(0, np.uint32)
    cnt = torch.tensor([len(words)], dtype=torch.int64)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, cnt)                           # 1. the counts
    counts = [int(c.item()) for c in counts]
    pad = max(max(counts), 1)
    send = torch.zeros(pad, dtype=torch.int32)
    send[:len(words)] = torch.from_numpy(words.view(np.int32))
    recv = [torch.zeros(pad, dtype=torch.int32) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)                         # 2. the words (padded here: gloo has no gatherv either)
    ok = True
    if rank == 0:
        buf = np.concatenate([recv[r].numpy().view(np.uint32)[:counts[r]] for r in range(world)])
        msgs = loamx.dist_split_messages(buf, counts)
        got = [None if m is None else loamx.dist_unpack_clouds(m) for m in msgs]
        q.put(("root", counts, [[(p.tolist(), c.tolist(), s.tolist()) for (p, c, s) in g] if g is not None else None for g in got]))
    q.put((rank, [(poses[s].tolist(), corners[s].tolist(), surfs[s].tolist()) for s in range(n_streams)]))
    dist.barrier()
    dist.destroy_process_group()
    return ok


def test_epoch_cloud_messages_world2():
    """SURVEY.md section 8e, collective 3: variable-size clouds to the accumulator's rank — layout and count-first protocol on CPU"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cloud_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    items = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    root = next(i for i in items if i[0] == "root")
    sent = {i[0]: i[1] for i in items if i[0] != "root"}
    assert root[1][1] == 0 and root[1][0] > 0                  # rank 1's empty message
    assert root[2][1] is None
    assert len(root[2][0]) == 3
    for s_, (pose, corner, surf) in enumerate(root[2][0]):      # bit for bit what rank 0 packed
        assert pose == sent[0][s_][0] and corner == sent[0][s_][1] and surf == sent[0][s_][2]
    assert root[2][0][0][1] == []                               # the empty cloud came through as empty


def test_cloud_message_is_validated():
    from loam_velodyne_amd import loamx
    w = loamx.dist_pack_clouds([np.ones((3, 4), np.float32)], [np.ones((5, 4), np.float32)], np.zeros((1, 6), np.float32))
    assert len(w) == 2 + 2 + 6 + 4 * 8
    assert len(loamx.dist_unpack_clouds(w)) == 1
    for bad in (w[:-1], np.concatenate([w, w[:1]]), np.concatenate([[1], w[1:]]).astype(np.uint32)):
        with pytest.raises(loamx.LoamxError):
            loamx.dist_unpack_clouds(bad)
    # PCL-layout clouds (32-byte records) pack to the same words
    a, b = np.arange(12, dtype=np.float32).reshape(3, 4), np.arange(20, dtype=np.float32).reshape(5, 4)
    w1 = loamx.dist_pack_clouds([a], [b], np.zeros((1, 6), np.float32))
    w2 = loamx.dist_pack_clouds([loamx.to_pcl_layout(a)], [loamx.to_pcl_layout(b)], np.zeros((1, 6), np.float32))
    assert np.array_equal(w1, w2)
