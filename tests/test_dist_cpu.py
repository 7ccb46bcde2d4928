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


def test_stream_layout_helpers():
    from loam_velodyne_amd import dist as lxdist
    all_ids = sum((lxdist.stream_ids(r, 4, 8) for r in range(4)), [])
    assert all_ids == list(range(32))                                  # BASELINE configs[3]: batch 32 over 4 GPUs
    starts = {lxdist.stream_start(g) for g in range(64)}
    assert len(starts) == 64
    assert lxdist.split_map(1_000_000) == (100_000, 900_000)
