"""Rank body for tests/test_gpu_dist.py and tests/test_launch_cpu.py (started by loam_velodyne_amd.launch).
argv[1] = output directory; argv[2] = "fake" (CPU: the rendezvous protocol only, RCCL replaced by a recorder) or "gpu"."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loam_velodyne_amd import launch, loamx  # noqa: E402

out_dir, mode = sys.argv[1], sys.argv[2]

if mode == "fake":
    class FakeDist:                       # stands in for the RCCL communicator: records what rendezvous() hands over
        @staticmethod
        def unique_id():
            return bytes([7, 1, 2, 3] * 32)

        def __init__(self, uid, rank, world, device):
            self.uid, self.rank, self.world, self.device = uid, rank, world, device
    loamx.Dist = FakeDist
    d, rank, world, local = launch.rendezvous(timeout_s=30)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "local": local, "uid": d.uid.hex(), "device": d.device}, f)
    sys.exit(0)

# ---- gpu: every rank registers its shard of a batch against the map broadcast from rank 0
import torch  # noqa: E402  (device memory only)
from loam_velodyne_amd import synth  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as op  # noqa: E402

d, rank, world, local = launch.rendezvous()
torch.cuda.set_device(local)
B = int(os.environ.get("LOAMX_TEST_BATCH", "4"))   # (5 with two ranks: shards of 2 and 3 sweeps)
w = synth.World(half_extent=65.0)
n_corner, n_surf = 10000, 90000
map_t = torch.zeros((n_corner + n_surf, 4), dtype=torch.float32, device=f"cuda:{local}")
if rank == 0:
    cm, sm = w.make_map(n_corner + n_surf)
    map_t.copy_(torch.from_numpy(np.concatenate([cm, sm])))
torch.cuda.synchronize()
ev = d.broadcast_map(map_t.data_ptr(), n_corner, map_t.data_ptr() + 16 * n_corner, n_surf, root=0)
orc = op.Oracle()
sr = op.ScanRegistration(orc)
rng = np.random.default_rng(1)
cl, sl, guesses = [], [], []
for k in range(B):                         # every rank generates the whole batch (deterministic), registers its shard
    gt = np.array([0.01 * rng.normal(), 0.3 * rng.normal(), 0.01 * rng.normal(), 3 * rng.normal(), 0.05 * rng.normal(), 3 * rng.normal()])
    sw = synth.make_sweep(w, "VLP-16", gt, gt, seed=100 + k, az_steps=900)
    f = sr.process(sw.points, sw.ring_sizes)
    c, s = f["less_sharp"].copy(), f["less_flat"].copy()
    c[:, 3] = np.floor(c[:, 3]); s[:, 3] = np.floor(s[:, 3])
    cl.append(c); sl.append(s)
    guesses.append(gt + np.array([0.003, 0.003, 0.003, 0.05, 0.05, 0.05]) * rng.normal(size=6))
guesses = np.array(guesses, np.float32)
b0, b1 = d.shard(B)
bt = loamx.Batch(max(b1 - b0, 1), device=local)
bt.stage_frozen_device(map_t.data_ptr(), n_corner, map_t.data_ptr() + 16 * n_corner, n_surf, wait_event=ev)   # index build ordered behind the broadcast
bt.swap_frozen()
if b1 > b0:
    bt.upload(cl[b0:b1], sl[b0:b1], guesses[b0:b1])
    bt.run()
    poses, stats = bt.download()
else:                                      # more ranks than sweeps: an empty shard still takes part in the exchanges
    poses, stats = np.zeros((0, 6), np.float32), np.zeros((0, 4), np.int32)
allp, allf, counts = d.allgather_results(poses, np.stack([stats[:, 0], stats[:, 1]], 1), batch=B)
assert int(counts.sum()) == B and d.comm_count() == world
# the epoch's merge exchange: every rank's (variable-size) cloud message reaches rank 0 intact (loamx_dist_gatherv)
rngm = np.random.default_rng(7 + rank)
mine = [(rngm.normal(size=6).astype(np.float32), rngm.normal(size=(30 + 11 * rank + k, 4)).astype(np.float32), rngm.normal(size=(80 + k, 4)).astype(np.float32))
        for k in range(1 + rank)]
words = loamx.dist_pack_clouds([m[1] for m in mine], [m[2] for m in mine], [m[0] for m in mine])
msgs, wcounts = d.gatherv(words, root=0)
assert int(wcounts[rank]) == len(words)
if rank == 0:
    assert len(msgs) == world
    for r_, m_ in enumerate(msgs):
        rr = np.random.default_rng(7 + r_)
        for k, (pose, co, su) in enumerate(loamx.dist_unpack_clouds(m_)):
            assert np.array_equal(pose, rr.normal(size=6).astype(np.float32))
            assert np.array_equal(co, rr.normal(size=(30 + 11 * r_ + k, 4)).astype(np.float32)) and np.array_equal(su, rr.normal(size=(80 + k, 4)).astype(np.float32))
else:
    assert msgs is None
d.barrier()
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), poses=allp, flags=allf, shard=np.array([b0, b1]), map_sum=float(map_t.double().sum().item()))
