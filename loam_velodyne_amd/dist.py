"""Multi-GPU plumbing of the batched-sweep mode (harness side; torch.distributed is transport only).

The path shards by independent streams: rank r owns streams [r*S, (r+1)*S) — no data-path collective.
The only exchange is the frozen map: one broadcast from rank 0 per map epoch (RCCL over xGMI with backend "nccl";
gloo on CPU for the tests), after which every rank builds its own spatial index.
"""
from __future__ import annotations

import numpy as np


def stream_ids(rank: int, world: int, streams_per_rank: int):
    """Global stream ids owned by `rank` (contiguous block)."""
    assert 0 <= rank < world
    return list(range(rank * streams_per_rank, (rank + 1) * streams_per_rank))


def stream_start(gs: int):
    """Deterministic start position of global stream `gs` inside the synthetic hall (x, y, z)."""
    return (3.0 * (gs % 8) - 10.0, 0.0, -40.0 + 9.0 * ((gs // 8) % 8) + 2.0 * (gs % 3))


def split_map(n_points: int, corner_fraction: float = 0.1):
    n_corner = int(round(n_points * corner_fraction))
    return n_corner, n_points - n_corner


def broadcast_map(map_tensor, dist, src: int = 0):
    """Ship the (M,4) float32 map tensor from `src` to every rank, in place.  Returns the tensor."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(map_tensor, src=src)
    return map_tensor


def max_over_ranks(value: float, dist, device=None) -> float:
    """The job-level step time is the slowest rank's."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_poses(poses: np.ndarray, dist, device=None) -> np.ndarray:
    """All ranks' (S,6) poses stacked in global stream order (tiny all-gather; reporting only)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(poses, np.float32)
    import torch
    t = torch.from_numpy(np.ascontiguousarray(poses, np.float32))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out, 0).cpu().numpy()


def epoch_merge(streams, acc, ldist=None, rank: int = 0, root: int = 0):
    """The merge step of a map epoch (collective 3; BasicLaserMapping.cpp:536-593 is what loamx_map_insert restates).

    streams: this rank's [(pose6 = transformAftMapped, corner_last, surf_last)] — per mapped stream the re-projected clouds of its last
    step (loamx_pipeline_download_last_clouds).  Every rank packs them into ONE message (loamx_dist_pack_clouds); the messages travel to
    `root` counts first (ldist.gatherv: RCCL send / recv; ldist = None: a single rank, the message goes through pack / unpack all the
    same); the root unpacks every rank's streams in rank order and inserts each sweep into the accumulator `acc` (a loamx.LaserMapping
    loaded with the epoch's map) with ITS pose, as given.  Returns the number of sweeps merged on the root, 0 elsewhere.
    """
    from loam_velodyne_amd import loamx
    words = loamx.dist_pack_clouds([s[1] for s in streams], [s[2] for s in streams], [s[0] for s in streams]) if streams else np.zeros(0, np.uint32)
    if ldist is None:
        msgs = [words if len(words) else None]
    else:
        msgs, _ = ldist.gatherv(words, root=root)
    if rank != root or msgs is None:
        return 0
    merged = 0
    for m in msgs:
        if m is None:
            continue
        for pose, corner, surf in loamx.dist_unpack_clouds(m):
            if len(corner) or len(surf):
                acc.insert(corner, surf, pose)
                merged += 1
    return merged


class AsyncEpochMerger:
    """Collective 3 off the stepping thread: the accumulator — gather of the ranks' sweeps, insertion into the
    map (loamx_map_insert: BasicLaserMapping.cpp:536-593 restated), download of the merged cubes, upload + broadcast of the new epoch's
    map — runs on a worker thread of its own with a communicator of its own, while the stepping thread goes on registering against the
    map it has.  The stepping thread only
      * hands over its streams' latest sweeps when the worker is idle (submit), and
      * asks between two steps whether a merged map has arrived (take_ready) — then stages it (loamx_pipeline_stage_frozen_device: the
        index is built in the background) and swaps it in at the next epoch boundary.
    Which step adopts which merged map therefore depends on timing; the synchronous protocol (epoch_merge on the stepping thread at every
    boundary) is the deterministic one, and the one the tests pin.  Every rank runs one merger; their workers meet in the collectives
    (gatherv, counts, broadcast), so job i is the same job on every rank.

    publish(corner, surf) -> token: called on the worker thread after the merge; on the root with the merged cubes (elsewhere None, None):
    must make the new map resident on every rank (sizes exchange + broadcast on the WORKER's communicator) and return whatever the stepping
    thread needs to stage it (device pointers, sizes, the event to wait for)."""

    def __init__(self, acc, ldist, rank, root, publish, before_job=None):
        import queue
        import threading
        self.acc, self.ldist, self.rank, self.root, self.publish = acc, ldist, rank, root, publish
        self.before_job = before_job
        self.q = queue.Queue()
        self.lock = threading.Lock()
        self.ready = []
        self.busy = False
        self.error = None
        self.jobs_done = 0
        self.merged_sweeps = 0
        self.job_seconds = []
        self.th = threading.Thread(target=self._run, name="loamx-epoch-merger", daemon=True)
        self.th.start()

    def _run(self):
        import time
        while True:
            job = self.q.get()
            if job is None:
                return
            try:
                t0 = time.perf_counter()
                if self.before_job is not None:
                    self.before_job()
                merged = epoch_merge(job, self.acc, self.ldist, rank=self.rank, root=self.root)
                corner = surf = None
                if self.rank == self.root:
                    corner, surf = self.acc.cubes("corner"), self.acc.cubes("surf")
                token = self.publish(corner, surf)
                with self.lock:
                    self.ready.append(token)
                    self.jobs_done += 1
                    self.merged_sweeps += merged
                    self.job_seconds.append(time.perf_counter() - t0)
                    self.busy = False
            except BaseException as e:   # noqa: BLE001 (reported by the stepping thread)
                with self.lock:
                    self.error = e
                    self.busy = False
                return

    def idle(self) -> bool:
        with self.lock:
            if self.error is not None:
                raise self.error
            return not self.busy

    def submit(self, streams):
        with self.lock:
            self.busy = True
        self.q.put(streams)

    def take_ready(self):
        """the newest merged map that has arrived since the last call (older ones that were never staged are dropped), or None"""
        with self.lock:
            if self.error is not None:
                raise self.error
            if not self.ready:
                return None
            tok = self.ready[-1]
            self.ready.clear()
            return tok

    def close(self, wait=True):
        self.q.put(None)
        if wait:
            self.th.join(timeout=120)
        if self.error is not None:
            raise self.error
