#!/usr/bin/env python
"""bench.py — sweeps/sec of the registration hot path (feature extraction + odometry + scan-to-map registration).

Workload at every N (weak scaling, BASELINE.json configs[3] = "HDL-64 sweep, 1M-pt map, batch 32 over 4 GPUs", i.e.
8 sweeps in flight per GPU): each GPU runs 8 independent streams of synthetic HDL-64E sweeps (64 x 2048 = 131,072
points) against a frozen 1,000,000-point sub-map.  A "step" advances every stream by one sweep through the whole path:
  features (BasicScanRegistration) -> odometry (BasicLaserOdometry) -> registration (BasicLaserMapping, frozen map).
All sweeps are staged in HBM before the timed region; the map is generated on rank 0 and shipped to the other ranks
with one RCCL broadcast (torch.distributed, backend "nccl" = RCCL over xGMI) — the only collective on the path; the
streams themselves are sharded with no data-path exchange.

One JSON line on rank 0 (the driver's contract) plus `roofline` (dominant kernel: k_gn_iter — one fused Gauss-Newton
iteration: neighbour search + fit + normal equations; algorithmic bytes = 72 B x query-iterations, duration from HIP events
on the library's own stream) and `cpu_baseline` (the oracle = CPU restatement of the reference, -O3 -march=native, every
stage single-threaded, 20 measured sweeps of the same workload: median / p95, serial and 3-stage-pipelined rates, and the
reference's own translation units timed beside it).

`--mode live` measures BASELINE configs[1] instead: VLP-16 sweeps, 200 k-pt LIVE map, one sweep in flight through the
single-stream entry points (loamx_scanreg_process / loamx_odom_process / loamx_map_process: host clouds in and out, map
updated and re-voxelised every sweep, sub-map index rebuilt every sweep) — the sequential-SLAM mode.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STREAMS_PER_GPU = 8
TIMING_PERIOD = 10        # every 10th step of the timed region carries the HIP-event timing (stage chains, k_gn_iter / odometry launches); every 4th cost 2.4 % of the window, every 10th nothing measurable (profiles/r06_ab.md section 17)
HANDLES_PER_GPU = 1      # >1: split the streams over several pipeline handles driven from host threads (measured: no gain)
SENSOR = "HDL-64E"
MAP_POINTS = 1_000_000
HBM_PEAK_GBS = 8000.0
HBM_COPY_GBS = 6290.0      # measured float4 copy on MI355X (MI355X_MICROARCH.md): the ceiling a streaming kernel can reach, quoted beside the 8 TB/s peak
PROFILE_ROUNDS = ("r06", "r05", "r04")   # committed rocprofv3 summaries of this command, newest first (profiles/<round>_bench_kernel_stats.csv, <round>_pmc_summary.json)


def unbind_worker():
    """initializer of the sweep-generating worker processes: bench.py binds itself (and the pinned memory it first-touches) to the NUMA node of
    its GPU, and children inherit the mask — but ray casting is memory-bound numpy, and 32 workers on one socket's memory controllers took
    96 s for the long window's 3,296 sweeps (0.5 s per sweep on an idle core)"""
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except OSError:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--map-points", type=int, default=MAP_POINTS)
    ap.add_argument("--map-epoch-steps", type=int, default=0,
                    help="E > 0: a new map epoch every E steps inside the timed region — rank 0's map is re-broadcast (RCCL, async), "
                         "indexed in the background and swapped in at the epoch boundary (BASELINE configs[4] double buffering)")
    ap.add_argument("--epoch-merge", action="store_true",
                    help="with --map-epoch-steps: the epoch's merge step too — every rank's registered sweeps (re-projected clouds + mapped pose) travel to "
                         "rank 0 (loamx_dist_gatherv), are inserted into the map accumulator there (loamx_map_insert) and the MERGED map is what the "
                         "next epoch broadcasts; synchronous on the stepping thread (the accumulator's ~1-2 ms per sweep is rank 0's)")
    ap.add_argument("--epoch-merge-async", action="store_true",
                    help="with --map-epoch-steps: the merge step OFF the stepping thread — the accumulator (gather, loamx_map_insert, cubes download, upload + broadcast "
                         "of the merged map) runs on a worker thread with a communicator of its own (loam_velodyne_amd.dist.AsyncEpochMerger); the stepping thread hands "
                         "over its streams' latest sweeps whenever the worker is idle and stages a merged map when one has arrived (swapped in at the next boundary)")
    ap.add_argument("--sensor", default=SENSOR, choices=["HDL-32", "HDL-64E", "VLP-16"], help="parity / side configurations (BASELINE configs[1], [2])")
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU)
    ap.add_argument("--handles", type=int, default=HANDLES_PER_GPU,
                    help="pipeline handles per GPU, each on its own HIP stream and host thread, sharing the streams evenly")
    ap.add_argument("--no-pcie", action="store_true", help="skip the second, PCIe-inclusive timed window")
    ap.add_argument("--repeat", type=int, default=5, help="repeat the timed window this many times (fresh handles, same sweeps): value_median / value_min / value_max; `value` is the first window")
    ap.add_argument("--ab-pcie", default=None, help="diagnostic: like --ab, but each variant runs the PCIe-inclusive window")
    ap.add_argument("--ab", default=None, help="diagnostic: ';'-separated environment variants ('A=1 B=2;C=3;' — empty = defaults) timed inside this process before the contract's window, one line each on stderr")
    ap.add_argument("--long-steps", type=int, default=400,
                    help="N > 0 (default 400; single rank, with the CPU baseline only): an additional window of N steps over a trajectory of its own, "
                         "reported as value_long — ~0.2 s of device time, twice (timed, then once more for the iteration counts), so that an external "
                         "sampler (rocm-smi) can see the GPU busy; costs ~60 s of host time for the sweeps; 0 = skip")
    ap.add_argument("--no-live-nodes", action="store_true", help="live mode: skip the additional run with the three entry points as concurrent nodes")
    ap.add_argument("--no-envelope", action="store_true", help="skip the reference-envelope chains of the long window (five CPU worker processes, ~100 s beside the other blocks)")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the blocks of the other single-GPU configurations (live_vlp16, live_hdl32, map_2m)")
    ap.add_argument("--map2-points", type=int, default=2_000_000, help="size of the second frozen sub-map (the map_2m block: BASELINE configs[4]'s single-GPU point)")
    ap.add_argument("--live-steps", type=int, default=100, help="timed sweeps of each sequential-SLAM block (10 warm-up sweeps in front)")
    ap.add_argument("--mode", default="batched", choices=["batched", "live"],
                    help="live: sequential SLAM (BASELINE configs[1]: VLP-16, 200 k-pt live map, one sweep in flight, host clouds in / out)")
    args = ap.parse_args()
    if args.mode == "live":
        return run_live(args)

    # the pipeline keeps three HIP streams busy; with torch's and RCCL's streams in the same process the runtime's default of
    # 4 hardware queues can alias two of them onto one queue (measured -15 %): ask for 8 before the HIP runtime starts
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    numa_node = None if os.environ.get("LOAMX_BENCH_NO_BIND") else bind_near_gpu(torch, local_rank)   # (LOAMX_BENCH_NO_BIND: diagnostic)

    from loam_velodyne_amd import loamx, synth
    from loam_velodyne_amd import dist as lxdist

    ns, K, W = args.streams, args.steps, args.warmup
    T = 1 + W + K   # first sweep of a stream only initialises the odometry
    # The library's look-ahead runs the odometry (and the features) of up to LOOK steps beyond the step being registered.  A window that
    # ended with the last staged sweep would start with LOOK steps of look-ahead work already done and do none for the steps after it:
    # K registrations but only K - LOOK odometry passes inside the timed region.  So LOOK more steps are staged than are run, and the
    # window closes only when the look-ahead has drained (loamx_pipeline_drain_lookahead): what was done ahead before the window opens
    # is done ahead, for the steps after it, before it closes — K passes of every stage inside, a steady-state window.
    LOOK = int(os.environ.get("LOAMX_BENCH_LOOK", 6))   # = loamx_pipeline_lookahead_depth() of a staged batch (asserted when the pipeline exists)
    T_all = T + LOOK
    world_model = synth.World(half_extent=125.0)

    # ---- frozen map: generated on rank 0, broadcast over RCCL, adopted in place by the library
    M = args.map_points
    n_corner, n_surf = lxdist.split_map(M)
    map_t = torch.empty((M, 4), dtype=torch.float32, device=dev)
    map_host = None
    if rank == 0:
        cm, sm = world_model.make_map(M)
        assert len(cm) == n_corner and len(sm) == n_surf
        map_t.copy_(torch.from_numpy(np.concatenate([cm, sm], axis=0)))
        map_host = (cm, sm)
    t_bcast = 0.0
    bcast_via = "none (single rank)"
    ldist = None
    if dist is not None:
        # the map broadcast goes through the library's own RCCL communicator (loamx_dist_*, one ncclBroadcast per buffer); its
        # 128-byte unique id travels over the process group the driver's launcher already set up
        torch.cuda.synchronize()
        try:
            uid = torch.zeros(loamx.Dist.ID_BYTES, dtype=torch.uint8, device=dev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(loamx.Dist.unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, src=0)
            ldist = loamx.Dist(bytes(uid.cpu().numpy().tobytes()), rank, world, local_rank)
            ldist.broadcast_map(map_t.data_ptr(), n_corner, map_t.data_ptr() + 16 * n_corner, n_surf, root=0)   # warm-up (communicator set-up)
            ldist.barrier()
            t0 = time.perf_counter()
            ldist.broadcast_map(map_t.data_ptr(), n_corner, map_t.data_ptr() + 16 * n_corner, n_surf, root=0)
            ldist.barrier()
            t_bcast = time.perf_counter() - t0
            bcast_via = "loamx_dist_broadcast_map (RCCL, native)"
        except Exception as e:   # never lose the scaling run over the transport: fall back to the harness-side collective and say so
            ldist = None
            t0 = time.perf_counter()
            lxdist.broadcast_map(map_t, dist, src=0)
            torch.cuda.synchronize()
            t_bcast = time.perf_counter() - t0
            bcast_via = "torch.distributed broadcast (native path failed: %s)" % repr(e)[:160]

    # the asynchronous epoch merge's worker thread talks over a communicator of its own (an RCCL communicator serves one thread at a time)
    ldist_merge = None
    if dist is not None and ldist is not None and args.epoch_merge_async and args.map_epoch_steps > 0:
        uid2 = torch.zeros(loamx.Dist.ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid2.copy_(torch.frombuffer(bytearray(loamx.Dist.unique_id()), dtype=torch.uint8))
        dist.broadcast(uid2, src=0)
        ldist_merge = loamx.Dist(bytes(uid2.cpu().numpy().tobytes()), rank, world, local_rank)

    # ---- this rank's streams and staged sweeps (distinct trajectories per rank and stream)
    sweeps = [[None] * ns for _ in range(T_all)]
    starts = []
    jobs = []
    for s, gs in enumerate(lxdist.stream_ids(rank, world, ns)):
        start = lxdist.stream_start(gs)
        poses = synth.trajectory(T_all, start=start)
        starts.append(np.array([0, 0, 0, start[0], start[1], start[2]], np.float32))
        for t in range(T_all):
            jobs.append((t, s, (125.0, args.sensor, poses[t], poses[t + 1], 1000 * gs + t)))
    # ray casting is ~0.5 s per HDL-64E sweep on one core and the GPU box is leased by the minute: the (seeded, order-independent)
    # sweeps are generated by worker processes — spawned, not forked: the HIP runtime is already up in this one
    # (32: measured on the pool's 256-thread hosts — 96 workers took four times as long for the long window's 3,296 sweeps as 32 did)
    n_workers = max(1, min(int(os.environ.get("LOAMX_BENCH_WORKERS", "32")), len(os.sched_getaffinity(0)) // max(world, 1), len(jobs)))
    if n_workers > 1:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=n_workers, mp_context=mp.get_context("spawn"), initializer=unbind_worker) as ex:
            made = list(ex.map(synth.make_sweep_job, [j[2] for j in jobs], chunksize=max(1, len(jobs) // (4 * n_workers))))
    else:
        made = [synth.make_sweep_job(j[2]) for j in jobs]
    for (t, s, _), (pts, rs) in zip(jobs, made):
        sweeps[t][s] = (pts, rs)
    n_points = len(sweeps[0][0][0])

    from concurrent.futures import ThreadPoolExecutor
    E = args.map_epoch_steps

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def resident_window(keep_open=False, collect=None):
        """W warm-up steps, then exactly K timed steps bracketed by barrier + synchronize, all sweeps resident in HBM.
        collect: a list that receives, per step, the (odometry sum, mapped pose, iteration counts) of every stream — used by the
        separate, untimed parity window only (it costs a few microseconds per stream and step).
        With H > 1 pipeline handles (--handles / LOAMX_BENCH_HANDLES) the streams are dealt over H handles, each driven by a host thread
        of its own that runs ITS K steps without waiting for the others between steps (the streams are independent; only the window's
        two ends are common)."""
        H = max(1, min(int(os.environ.get("LOAMX_BENCH_HANDLES", args.handles)), ns))
        assert ns % H == 0, "--streams must be a multiple of --handles"
        assert E == 0 or H == 1, "--map-epoch-steps runs with one handle"
        per = ns // H
        torch.cuda.synchronize()
        pipes = []
        for h in range(H):
            p = loamx.Pipeline(per, scanreg=dict(device=local_rank), odom=dict(device=local_rank), mapping=dict(device=local_rank))
            p.set_frozen_device(map_t.data_ptr(), n_corner, map_t.data_ptr() + 16 * n_corner, n_surf)
            for k in range(per):
                p.set_state(k, aft=starts[h * per + k])
            p.upload([[sweeps[t][h * per + k] for k in range(per)] for t in range(T_all)])
            p.set_timing(True)
            assert p.lookahead_depth() <= LOOK, "the window stages fewer steps beyond its last one than the look-ahead runs ahead"
            pipes.append(p)
        r = dict(stage=np.zeros(4), res_ms=0.0, res_launches=0, q_iters=0, queries=0, n_sampled=0, in_step=0.0, n_epochs=0, handles=H, per=per)
        racc = [dict(stage=np.zeros(4), res_ms=0.0, res_launches=0, q_iters=0, queries=0, n_sampled=0, in_step=0.0) for _ in range(H)]
        # double-buffered map epochs (off by default): epoch k+1 is broadcast and indexed in the background during epoch k
        # and swapped in before the first step of epoch k+1
        map_nexts = [torch.empty_like(map_t), torch.empty_like(map_t)] if E > 0 else None   # ping-pong: a buffer is rewritten only
        # after a registration against the index built from it has been observed complete
        ev_map = torch.cuda.Event() if E > 0 else None
        # the epoch's merge step (--epoch-merge): rank 0 owns the accumulator, loaded with the first epoch's map
        acc = None
        if E > 0 and (args.epoch_merge or args.epoch_merge_async) and rank == 0:
            acc = loamx.LaserMapping(device=local_rank)
            acc.load_cubes(*map_host)
        r["merged_sweeps"] = 0
        merger = None
        if E > 0 and args.epoch_merge_async:
            # three device buffers in rotation: the map being registered against, the one staged behind it, the one being filled
            bufs = [torch.empty((int(1.3 * M) + 65536, 4), dtype=torch.float32, device=dev) for _ in range(3)]
            turn = [0]
            ev_pub = [torch.cuda.Event() for _ in range(3)]

            def publish(corner, surf):   # (worker thread)
                b = turn[0] % 3
                turn[0] += 1
                nc_e, ns_e = (len(corner), len(surf)) if corner is not None else (0, 0)
                if ldist_merge is not None:   # the new sizes reach every rank the way the counts of the other exchanges do
                    nc_e = int(ldist_merge.allgather_counts(nc_e)[0]); ns_e = int(ldist_merge.allgather_counts(ns_e)[0])
                if bufs[b].shape[0] < nc_e + ns_e:
                    bufs[b] = torch.empty((int(1.2 * (nc_e + ns_e)), 4), dtype=torch.float32, device=dev)
                if corner is not None:
                    bufs[b][:nc_e + ns_e].copy_(torch.from_numpy(np.concatenate([corner, surf], axis=0)))
                ev_pub[b].record()
                ev = ev_pub[b].cuda_event
                if ldist_merge is not None:
                    ev = ldist_merge.broadcast_map(bufs[b].data_ptr(), nc_e, bufs[b].data_ptr() + 16 * nc_e, ns_e, root=0, wait_event=ev)
                return (bufs[b].data_ptr(), nc_e, bufs[b].data_ptr() + 16 * nc_e, ns_e, ev)

            merger = lxdist.AsyncEpochMerger(acc, ldist_merge, rank, 0, publish, before_job=lambda: torch.cuda.set_device(local_rank))
        r["merge_jobs"] = 0
        sizes = [n_corner, n_surf]   # of the map the NEXT epoch registers against (changes once sweeps are merged)

        def snapshot(h, t):
            if collect is not None:
                p = pipes[h]
                for k in range(per):
                    _, ts_, aft_, st_ = p.get(k)
                    collect.append((t, h * per + k, ts_.copy(), aft_.copy(), st_["odom_iterations"], st_["map_iterations"],
                                    (st_["odom_sel"], st_["map_sel"], st_["corner_ds"], st_["surf_ds"])))

        def warm(h):   # warm-up (includes every stream's initialising first sweep); the window opens with the look-ahead exactly
            p = pipes[h]                 # LOOK steps ahead — and closes the same way (below)
            for t in range(1 + W):
                p.step(t)
                snapshot(h, t)
            p.drain_lookahead()

        period = int(os.environ.get("LOAMX_BENCH_TIMING_PERIOD", TIMING_PERIOD))   # (diagnostic override: what the sampled steps cost)

        def timed(h):
            p, a = pipes[h], racc[h]
            for t in range(1 + W, T):
                if E > 0:
                    k = (t - (1 + W)) % E
                    if k == 0:
                        if p.swap_frozen():
                            r["n_epochs"] += 1
                        slot = ((t - (1 + W)) // E) % 2
                        map_next = map_nexts[slot]
                        nc_e, ns_e = sizes
                        if merger is not None:
                            # the accumulator runs beside the steps: hand it this rank's latest sweeps whenever it is idle, nothing else
                            # happens at a boundary (a merged map that has arrived was staged between two steps, below)
                            if t > 1 + W and merger.idle():
                                mine = []
                                for k_ in range(per):
                                    _, _, aft_, st_ = p.get(k_)
                                    if st_["mapped"]:
                                        lc_, ls_ = p.last_clouds(k_, n_points)
                                        mine.append((aft_, lc_, ls_))
                                merger.submit(mine)
                            k = -1   # (skip the synchronous staging below)
                    if k == 0:
                        if args.epoch_merge and t > 1 + W:
                            # collective 3: this rank's streams' last sweeps -> rank 0's accumulator -> the merged map
                            mine = []
                            for k_ in range(per):
                                _, _, aft_, st_ = p.get(k_)
                                if st_["mapped"]:
                                    lc_, ls_ = p.last_clouds(k_, n_points)
                                    mine.append((aft_, lc_, ls_))
                            r["merged_sweeps"] += lxdist.epoch_merge(mine, acc, ldist, rank=rank, root=0)
                            if rank == 0:
                                new_c, new_s = acc.cubes("corner"), acc.cubes("surf")
                                nc_e, ns_e = len(new_c), len(new_s)
                            if ldist is not None:   # the new sizes reach every rank the way the counts of the other exchanges do
                                nc_e = int(ldist.allgather_counts(nc_e)[0]); ns_e = int(ldist.allgather_counts(ns_e)[0])
                            if map_next.shape[0] < nc_e + ns_e:
                                map_next = map_nexts[slot] = torch.empty((int(1.2 * (nc_e + ns_e)), 4), dtype=torch.float32, device=dev)
                            if rank == 0:
                                map_next[:nc_e + ns_e].copy_(torch.from_numpy(np.concatenate([new_c, new_s], axis=0)), non_blocking=False)
                            sizes[0], sizes[1] = nc_e, ns_e
                        elif rank == 0:
                            map_next[:nc_e + ns_e].copy_(map_t[:nc_e + ns_e], non_blocking=True)   # (the next epoch's map: same content, new buffer)
                        if ldist is not None:   # native: the broadcast waits for the copy's event, the index build for the broadcast's
                            ev_map.record()
                            ev = ldist.broadcast_map(map_next.data_ptr(), nc_e, map_next.data_ptr() + 16 * nc_e, ns_e, root=0, wait_event=ev_map.cuda_event)
                        else:
                            if dist is not None:
                                dist.broadcast(map_next, src=0, async_op=True).wait()   # orders torch's stream behind RCCL's, not the host
                            ev_map.record()
                            ev = ev_map.cuda_event
                        # the index build waits for the event on the device; nothing blocks here
                        p.stage_frozen_device(map_next.data_ptr(), nc_e, map_next.data_ptr() + 16 * nc_e, ns_e, ev)
                if merger is not None:
                    tok = merger.take_ready()
                    if tok is not None:   # a merged map has arrived: index it in the background now, swap it in at the next boundary
                        p.stage_frozen_device(*tok)
                # HIP-event timing (stage chains + a pair around every Gauss-Newton / odometry launch) on every TIMING_PERIOD-th step only:
                # a sampled step is ~10 % slower (event records between dependent launches, their read-back)
                sampled = (t - (1 + W)) % period == 0
                p.set_timing(sampled)
                tc0 = time.perf_counter()
                p.step(t)
                a["in_step"] += time.perf_counter() - tc0
                snapshot(h, t)
                if sampled:   # event read-back of the step that just finished (the step itself is synchronous)
                    a["n_sampled"] += 1
                    tm = p.timing()
                    a["stage"] += np.array([tm["features_ms"], tm["odometry_ms"], tm["registration_ms"], tm["step_ms"]])
                    a["res_ms"] += tm["residual_ms"]
                    a["res_launches"] += tm["residual_launches"]
                    a["q_iters"] += tm["query_iterations"]
                    a["queries"] += tm["queries"]
            p.drain_lookahead()   # the look-ahead work for the LOOK steps after the window belongs inside it (see LOOK above)

        def on_all(fn):
            if H == 1:
                fn(0)
            else:   # ctypes releases the GIL inside the library: the handles really run concurrently
                for f in [pool.submit(fn, h) for h in range(H)]:
                    f.result()

        pool = ThreadPoolExecutor(max_workers=H) if H > 1 else None
        on_all(warm)
        sync_all()
        if E > 0:   # one untimed stage + swap so that the second set of index buffers exists before the timed region
            for p in pipes:
                p.stage_frozen_device(map_t.data_ptr(), n_corner, map_t.data_ptr() + 16 * n_corner, n_surf)
            torch.cuda.synchronize()
            for p in pipes:
                p.swap_frozen()
            sync_all()
        import gc
        gc.collect()
        gc.disable()   # (as timeit does: a collection over this process's heap — torch is loaded — pauses the harness for tens of ms, a 9 ms window cannot absorb that)
        t0 = time.perf_counter()
        on_all(timed)
        sync_all()
        r["elapsed"] = lxdist.max_over_ranks(time.perf_counter() - t0, dist, dev)
        gc.enable()
        if merger is not None:   # (the job in flight when the window closed finishes outside it)
            r["merge_jobs"], r["merged_sweeps"] = merger.jobs_done, merger.merged_sweeps
            r["merge_job_seconds"] = [round(x, 4) for x in merger.job_seconds]
            merger.close()
        if pool is not None:
            pool.shutdown()
        for a in racc:   # stage times: mean over the handles (they run side by side); counts: summed
            r["stage"] += a["stage"] / H
            for k in ("res_ms", "res_launches", "q_iters", "queries"):
                r[k] += a[k]
        r["n_sampled"] = racc[0]["n_sampled"]
        r["in_step"] = max(a["in_step"] for a in racc)
        # the odometry chains' launch pairs (HIP events on the chains' own streams on the sampled steps; totals since the handles were
        # created, i.e. incl. the warm-up's sampled steps — the same kernels on the same sweeps)
        r["odom_launch"] = {}
        for p in pipes:
            for k_, v_ in p.odom_launch_timing().items():
                r["odom_launch"][k_] = r["odom_launch"].get(k_, 0) + v_
        r["pipes"] = pipes
        if not keep_open:
            for p in pipes:
                p.close()
            r["pipes"] = None
        return r

    def long_window(N):
        """value_long: the same protocol over N timed steps — ~0.2 s of device time, long enough for an external sampler to see the GPU
        busy.  Its sweeps are generated for it: the short window's trajectory (1 m and 0.5 degrees per sweep, a circle of 115 m radius)
        leaves the 250 m world after ~100 sweeps, so the long one turns tighter (1.43 degrees per sweep: a 40 m circle that stays inside
        the map).  The iteration counts of the long window are reported with it — they, not the window length, are what makes its
        sweeps cheaper or dearer than the short window's."""
        nonlocal sweeps, T, T_all, K
        T_l = 1 + W + N + LOOK
        t_l0 = time.perf_counter()
        secs = {}
        jobs_l = []
        for s_, gs in enumerate(lxdist.stream_ids(rank, world, ns)):
            poses = synth.trajectory(T_l, yaw_step_deg=1.43, start=lxdist.stream_start(gs))
            for t in range(T_l):
                jobs_l.append((t, s_, (125.0, args.sensor, poses[t], poses[t + 1], 1000 * gs + t)))
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        jobs_l.sort(key=lambda j: (j[1] != 0, j[1], j[0]))   # stream 0 first: its sweeps are all the envelope chains need
        n0 = sum(1 for j in jobs_l if j[1] == 0)
        env_jobs = None
        m_ = map_t.cpu().numpy()
        with ProcessPoolExecutor(max_workers=n_workers, mp_context=mp.get_context("spawn"), initializer=unbind_worker) as ex:
            made_l = list(ex.map(synth.make_sweep_job, [j[2] for j in jobs_l[:n0]], chunksize=max(1, n0 // (4 * n_workers))))
            # the reference's own envelope over stream 0's sweeps: five worker processes, ~100 s — started NOW, beside the generation of the
            # other streams' sweeps (CPU work anyway), stopped (SIGSTOP) while the device windows below are timed, collected when the line
            # is assembled
            if not args.no_envelope and not args.no_cpu_baseline:
                try:
                    env_jobs = EnvelopeJobs("long", "frozen", np.stack([made_l[t][0] for t in range(1 + W + N)]), made_l[0][1],
                                            m_[:n_corner], m_[n_corner:], starts[0], 1 + W + N,
                                            ["oracle_fast", "ref", "ref_map_alt", "ref_odom_alt", "ref_both_alt"])
                except Exception as e:   # noqa: BLE001
                    env_jobs = None
                    print("bench.py: envelope chains not started: %r" % e, file=sys.stderr, flush=True)
            made_l += list(ex.map(synth.make_sweep_job, [j[2] for j in jobs_l[n0:]], chunksize=max(1, (len(jobs_l) - n0) // (4 * n_workers))))
        secs["sweeps_generated"] = round(time.perf_counter() - t_l0, 1)
        keep = (sweeps, T, T_all, K)
        sweeps = [[None] * ns for _ in range(T_l)]
        for (t, s_, _), (pts, rs) in zip(jobs_l, made_l):
            sweeps[t][s_] = (pts, rs)
        T, T_all, K = 1 + W + N, T_l, N
        try:
            if env_jobs is not None:
                env_jobs.pause()
            w_ = resident_window()
            per_step = []
            resident_window(collect=per_step)   # (untimed: the poses and iteration counts of every stream and step)
            sweeps_long = sweeps
            secs["two_windows"] = round(time.perf_counter() - t_l0 - secs["sweeps_generated"], 1)
        finally:
            if env_jobs is not None:
                env_jobs.resume()
            sweeps, T, T_all, K = keep
        timed_rows = [r for r in per_step if r[0] >= 1 + W]
        parity = None
        try:   # the oracle chain over the long trajectory's stream 0 (~0.1 s per sweep on one core): parity over hundreds of sweeps
            inputs = []
            chain = oracle_parity_chain(sweeps_long, starts, m_, n_corner, 1 + W + N, inputs=inputs)
            secs["oracle_chain"] = round(time.perf_counter() - t_l0 - secs["sweeps_generated"] - secs.get("two_windows", 0), 1)
            ps = per_step_check(loamx, m_[:n_corner], m_[n_corner:], inputs, chain)
            del inputs
            parity = ("pending", per_step, chain, ps)   # (completed by finish_long once the envelope chains are in)
        except Exception as e:
            parity = {"error": repr(e)[:200]}
        blk = {"value": round(world * ns * N / w_["elapsed"], 2), "unit": "sweeps/s", "steps": N, "ms_per_step": round(w_["elapsed"] / N * 1e3, 4),
               "pose_err_vs_oracle": parity,
               "seconds": round(w_["elapsed"], 4), "host_seconds": secs,
               "mean_odom_iterations": round(float(np.mean([r[4] for r in timed_rows])), 2), "mean_map_iterations": round(float(np.mean([r[5] for r in timed_rows])), 2),
               "note": "same window protocol as `value` over a longer trajectory of its own (40 m circle inside the map; the short window's 115 m "
                       "circle leaves the synthetic world after ~100 sweeps)"}
        return blk, env_jobs

    def finish_long(blk, env_jobs):
        """the long window's parity block, once the envelope chains have finished"""
        pr = blk.get("pose_err_vs_oracle")
        if not (isinstance(pr, tuple) and pr[0] == "pending"):
            if env_jobs is not None:
                env_jobs.collect()
            return
        _, per_step, chain, ps = pr
        env = None
        if env_jobs is not None:
            chains_ = env_jobs.collect()
            orc_rows = np.array([np.concatenate([[c[0]], c[1].astype(np.float64), c[2].astype(np.float64)]) for c in chain])
            env = reference_envelope(chains_, orc_rows)
            env["seconds_beside_the_other_blocks"] = round(env_jobs.seconds, 1)
        parity = pose_error(per_step, chain, stream=0, envelope=env, per_step=ps)
        blk["pose_err_vs_oracle"] = parity

    def map_2m_block():
        """BASELINE configs[4]'s single-GPU point: the same sweeps and streams against a 2,000,000-point frozen sub-map (twice the density
        over the same world) — the resident window once timed, once more for the poses, the oracle chain of stream 0 against the same map."""
        nonlocal map_t, n_corner, n_surf, M
        keep = (map_t, n_corner, n_surf, M)
        M2 = args.map2_points
        cm2, sm2 = world_model.make_map(M2)
        try:
            M = M2
            n_corner, n_surf = len(cm2), len(sm2)
            map_t = torch.empty((M2, 4), dtype=torch.float32, device=dev)
            map_t.copy_(torch.from_numpy(np.concatenate([cm2, sm2], axis=0)))
            torch.cuda.synchronize()
            w2 = resident_window()
            reps = [world * ns * K / w2["elapsed"]] + [world * ns * K / resident_window()["elapsed"] for _ in range(2)]
            blk = {"value": round(reps[0], 2), "unit": "sweeps/s", "ms_per_step": round(w2["elapsed"] / K * 1e3, 4), "steps": K, "warmup": W,
                   "value_median": round(float(np.median(reps)), 2), "value_repeats": len(reps),
                   "workload": f"BASELINE configs[4], one GPU: {args.sensor} sweeps ({n_points} pts), {M2}-pt frozen sub-map, {ns} streams, full path per sweep, resident"}
            S2 = max(w2["n_sampled"], 1)
            blk["stage_ms_per_step"] = {"features": round(w2["stage"][0] / S2, 4), "odometry": round(w2["stage"][1] / S2, 4), "registration": round(w2["stage"][2] / S2, 4)}
            if w2["res_launches"]:
                us = w2["res_ms"] / w2["res_launches"] * 1e3
                ach = 72.0 * w2["q_iters"] / w2["res_launches"] / (us * 1e-6) / 1e9
                blk["roofline"] = {"kernel": "loamx::k_gn_iter", "bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
                                   "avg_launch_us": round(us, 3), "launches": int(w2["res_launches"]), "algorithmic_bytes_per_launch": round(72.0 * w2["q_iters"] / w2["res_launches"], 1),
                                   "traffic": None, "model": "72 B per query-iteration (SURVEY.md §8d), HIP-event pairs on the sampled steps"}
            if not args.no_cpu_baseline:
                gp2, op2 = [], []
                resident_window(collect=gp2)
                cb = cpu_baseline(sweeps, starts, map_t, n_measure=8, n_warm=2, n_reference=0, poses_out=op2)
                blk["cpu_baseline"] = {k_: cb[k_] for k_ in ("value", "unit", "cores", "kind", "sample", "pipelined_value", "kdtree_build_seconds")}
                blk["pose_err_vs_oracle"] = pose_error(gp2, op2, stream=0)
            return blk
        finally:
            map_t, n_corner, n_surf, M = keep

    # ---- diagnostic: environment variants inside one process (same data, same box): --ab "A=1;B=2 C=3;"
    if args.ab is not None and world == 1:
        for v in args.ab.split(";"):
            kv = dict(x.split("=", 1) for x in v.split()) if v.strip() else {}
            saved = {k: os.environ.get(k) for k in kv}
            os.environ.update(kv)
            vals = []
            for _ in range(max(args.repeat, 1)):
                w_ = resident_window()
                vals.append(world * ns * K / w_["elapsed"])
            st_ = w_["stage"] / max(w_["n_sampled"], 1)
            print(f"[ab] {v.strip() or 'defaults':48s} sweeps/s median {np.median(vals):9.1f} min {min(vals):9.1f} max {max(vals):9.1f}  F {st_[0]:.3f} O {st_[1]:.3f} M {st_[2]:.3f}  "
                  f"gn {w_['res_ms'] / max(w_['res_launches'], 1) * 1e3:.1f} us x {w_['res_launches']}", file=sys.stderr, flush=True)
            for k, o in saved.items():
                if o is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = o

    if args.ab_pcie is not None and world == 1:
        for v in args.ab_pcie.split(";"):
            kv = dict(x.split("=", 1) for x in v.split()) if v.strip() else {}
            saved = {k: os.environ.get(k) for k in kv}
            os.environ.update(kv)
            rs = [pcie_inclusive_run(torch, loamx, local_rank, map_t, n_corner, n_surf, sweeps, starts, ns, W, K, dist, dev, lxdist) for _ in range(3)]
            print(f"[ab-pcie] {v.strip() or 'defaults':48s} sweeps/s " + " ".join(f"{r['value']:9.1f}" for r in rs) + f"  inside_step {rs[-1]['host_ms_per_step']['inside_step']:.3f} ms",
                  file=sys.stderr, flush=True)
            for k, o in saved.items():
                if o is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = o

    # ---- the contract's window: W warm-up steps, then exactly K timed steps
    win = resident_window(keep_open=True)
    pipes = win["pipes"]
    H, per = win["handles"], win["per"]
    elapsed = win["elapsed"]
    stage, res_ms, res_launches, q_iters_timed, queries_timed, n_sampled, in_step, n_epochs = (
        win["stage"], win["res_ms"], win["res_launches"], win["q_iters"], win["queries"], win["n_sampled"], win["in_step"], win["n_epochs"])

    # pose sanity of this rank's streams against ground truth (not the parity check — that is tests/)
    stats = [p.get(k)[3] for p in pipes for k in range(per)]
    n_results = ns
    if ldist is not None:   # every rank ends up with every stream's final pose (ncclAllGather of 8 x 4 B per stream)
        aft = np.stack([p.get(k)[2] for p in pipes for k in range(per)])
        allp, _, _ = ldist.allgather_results(aft, np.array([[st["map_iterations"], st["mapped"]] for st in stats], np.int32), batch=world * ns)
        n_results = len(allp)
    sweeps_total = world * ns * K
    value = sweeps_total / elapsed
    for p in pipes:   # free the window's HIP streams: beyond 8 streams per process the runtime aliases busy streams onto shared
        p.close()     # hardware queues and they serialise (GPU_MAX_HW_QUEUES above)
    # run-to-run spread: the same window (same sweeps, fresh handles) repeated; `value` stays the first window's
    repeats = [value]
    for _ in range(max(args.repeat, 1) - 1):
        repeats.append(sweeps_total / resident_window()["elapsed"])

    # ---- pose parity (BASELINE.json's metric, third part): one more window of the same steps, untimed, that records every stream's poses;
    # cpu_baseline() runs the oracle chain over stream 0's sweeps from the same start and the two are compared below (bar: 1e-4 m / rad)
    gpu_poses = []
    if world == 1 and not args.no_cpu_baseline:
        resident_window(collect=gpu_poses)

    # ---- the same K steps once more with the PCIe inside the timed region (SURVEY.md §8d "GPU timing"): every step's sweeps
    # are handed over from pinned host memory while earlier steps compute (loamx_pipeline_stage_step, seven steps ahead) and
    # every step's registered full-resolution clouds are copied back asynchronously (loamx_pipeline_download_step_async).  Never
    # `value`: reported beside it.
    pcie = None
    if H == 1 and not args.no_pcie:
        try:
            runs_ = [pcie_inclusive_run(torch, loamx, local_rank, map_t, n_corner, n_surf, sweeps, starts, ns, W, K, dist, dev, lxdist)
                     for _ in range(3 if args.repeat > 1 else 1)]
            runs_.sort(key=lambda r: r["value"])
            pcie = dict(runs_[len(runs_) // 2])   # the median window, the others listed beside it
            pcie["value_windows"] = [r["value"] for r in runs_]
        except Exception as e:   # the device-resident figure above is the contract; a failure here must not lose it
            pcie = {"error": repr(e)[:200]}


    if rank == 0:
        iters_map = np.mean([st["map_iterations"] for st in stats])
        iters_odom = np.mean([st["odom_iterations"] for st in stats])
        S = max(n_sampled, 1)   # steps that carried the event timing
        q_per_sweep = queries_timed / max(ns * S, 1)
        # BASELINE.md / SURVEY.md §8d algorithmic bytes per registered sweep (S = streams sharing the frozen map epoch)
        k_feat = 36 * synth.SENSORS[args.sensor][0]   # (2 sharp + 4 flat) x 6 regions per ring
        bytes_per_sweep = 32 * n_points + 16 * M / (ns * K) + 72 * (q_iters_timed / max(ns * S, 1)) + 48 * iters_odom * k_feat
        avg_launch_ms = res_ms / max(res_launches, 1)
        achieved = (72.0 * q_iters_timed / max(res_launches, 1)) / (avg_launch_ms * 1e-3) / 1e9 if res_launches else 0.0
        out = {
            "metric": "sweeps/sec (64-ring, 1M-pt map): feature extraction + odometry + scan-to-map registration",
            "value": round(value, 2),
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "value_median": round(float(np.median(repeats)), 2),
            "value_min": round(float(min(repeats)), 2),
            "value_max": round(float(max(repeats)), 2),
            "value_repeats": len(repeats),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.sensor} {synth.SENSORS[args.sensor][0]}x{synth.SENSORS[args.sensor][1]} sweeps ({n_points} pts), {M}-pt frozen sub-map, {ns} streams/GPU "
                            "(BASELINE configs[3]: batch 32 over 4 GPUs), full path per sweep",
                "streams_per_gpu": ns,
                "lookahead_depth": LOOK,
                "handles_per_gpu": H,
                "sweep_points": n_points,
                "map_points": M,
                "mean_map_iterations": round(float(iters_map), 2),
                "mean_odom_iterations": round(float(iters_odom), 2),
                "mean_queries_per_sweep": round(float(q_per_sweep), 1),
                "stage_ms_per_step": {"features": round(stage[0] / S, 4), "odometry": round(stage[1] / S, 4),
                                      "registration": round(stage[2] / S, 4), "gpu_step": round(stage[3] / S, 4)},
                "ms_per_step_inside_step_call": round(in_step / K * 1e3, 4),
                "stage_timing_sampling": f"HIP events on every {TIMING_PERIOD}th step of the timed region ({n_sampled} of {K} steps)",
                "timed_window": f"steady state: {LOOK} more steps are staged than run; the window opens and closes after "
                                "loamx_pipeline_drain_lookahead + synchronize, i.e. with the look-ahead exactly as far ahead at its end as at its "
                                f"start ({LOOK} steps): K passes of every stage inside the timed region",
                "map_broadcast_ms": round(t_bcast * 1e3, 3),
                "map_broadcast_via": bcast_via,
                "map_epoch_steps": E,
                "map_epochs_swapped": n_epochs,
                "map_epoch_merge": ({"merged_sweeps_on_rank0": win.get("merged_sweeps", 0),
                                     "note": "every epoch boundary: the ranks' registered sweeps -> loamx_dist_gatherv -> loamx_map_insert on rank 0 -> "
                                             "the merged map is broadcast and staged for the next epoch (synchronous, inside the timed region)"}
                                    if args.epoch_merge and E > 0 else
                                    {"mode": "asynchronous (loam_velodyne_amd.dist.AsyncEpochMerger)", "merge_jobs_completed_in_window": win.get("merge_jobs", 0),
                                     "merged_sweeps_on_rank0": win.get("merged_sweeps", 0), "job_seconds": win.get("merge_job_seconds"),
                                     "merged_maps_swapped_in": n_epochs,
                                     "note": "the accumulator runs on a worker thread beside the steps: it takes the ranks' latest sweeps whenever it is idle "
                                             "(gatherv -> loamx_map_insert on rank 0 -> cubes -> upload + broadcast on its own communicator); a merged map that has "
                                             "arrived is staged between two steps and swapped in at the next epoch boundary"}
                                    if args.epoch_merge_async and E > 0 else None),
                "numa_node_bound": numa_node,
                "results_gathered": n_results,
                # what the RCCL communicator itself reports (ncclCommCount); 0 = a multi-rank run that fell back to torch.distributed for
                # the map broadcast (map_broadcast_via says why): such a run must not be read as the native path
                "rccl_ranks": (ldist.comm_count() if ldist is not None else (1 if world == 1 else 0)),
                "path_algorithmic_bytes_per_sweep": round(float(bytes_per_sweep), 1),
                "path_hbm_frac": round(float(bytes_per_sweep * value / world / (HBM_PEAK_GBS * 1e9)), 6),
            },
            "pcie_inclusive": pcie,
            "roofline_gn": {
                "kernel": "loamx::k_gn_iter",
                "bound": "hbm",
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": pmc_traffic(),
                "traffic_note": "bytes of one launch with every sweep still iterating (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc "
                                f"passes of this command, read from the committed {pmc_source() or 'profiles/<round>_pmc_summary.json (none found)'}); achieved / avg_launch_us "
                                "average over all timed launches incl. the short ones after most sweeps have converged",
                "model": "72 B per query-iteration = 12 B query + 5 x 12 B neighbours (SURVEY.md §8d); the launch also fits edges / planes, "
                         "forms the 28 normal-equation sums and runs the 6x6 update step",
                "avg_launch_us": round(avg_launch_ms * 1e3, 3),
                "launches": res_launches,
                "launch_sampling": f"HIP-event pairs around every k_gn_iter launch on every {TIMING_PERIOD}th step of the timed region (a sampled step is ~10 % slower: event records between dependent launches)",
                "algorithmic_bytes_per_launch": round(72.0 * q_iters_timed / max(res_launches, 1), 1),
            },
        }
        t_phase = time.perf_counter()
        phases = {}

        def phase(name):
            nonlocal t_phase
            now = time.perf_counter()
            phases[name] = round(now - t_phase, 1)
            t_phase = now

        # ---- the other single-GPU configurations of BASELINE.json, each a block of this line (the driver runs this command only):
        # configs[4]'s one-GPU point (2 M-point map), configs[1] and configs[2] (sequential SLAM over a live map).  They run BEFORE the
        # CPU legs below: the long window's envelope chains occupy five cores for ~100 s, and the sequential-SLAM mode is bound by the
        # host's launch latency (measured with them beside it: VLP-16 1,290 instead of 1,470-1,500 sweeps/s)
        if world == 1 and not args.no_side_configs:
            try:
                out["map_2m"] = map_2m_block()
            except Exception as e:   # noqa: BLE001 (a side block must not lose the contract's line)
                out["map_2m"] = {"error": repr(e)[:300]}
            phase("map_2m")
            for key, sensor_, m_pts in (("live_vlp16", "VLP-16", 200_000), ("live_hdl32", "HDL-32", 500_000)):
                try:
                    out[key] = side_live(sensor_, m_pts, args.live_steps, 10, cpu=not args.no_cpu_baseline, nodes=not args.no_live_nodes)
                except Exception as e:   # noqa: BLE001
                    out[key] = {"error": repr(e)[:300]}
                phase(key)
        long_blk, env_jobs = None, None
        if world == 1 and not args.no_cpu_baseline:
            orc_poses = []
            out["cpu_baseline"] = cpu_baseline(sweeps, starts, map_t, poses_out=orc_poses)
            out["pose_err_vs_oracle"] = pose_error(gpu_poses, orc_poses, stream=0)
            phase("cpu_baseline")
            if args.long_steps:
                long_blk, env_jobs = long_window(args.long_steps)
                out["value_long"] = long_blk
                phase("long_window")
        if long_blk is not None:
            finish_long(long_blk, env_jobs)
            phase("envelope_wait")
        # `roofline` = the kernel with the largest total duration in the committed kernel stats of this command (its live figures measured
        # in THIS run); the Gauss-Newton kernel's block stays beside it
        out["roofline"] = dominant_roofline(out.pop("roofline_gn"), win.get("odom_launch") or {}, ns)
        out["roofline_kernels"] = roofline_kernels(out["roofline"], ns)
        out["config"]["env_overrides"] = env_overrides()
        out["config"]["library_build"] = loamx.build_info()
        out["config"]["bench_phase_seconds"] = phases
        # ---- every configuration's headline figures twice more: flat in `config` (scalars survive any record that keeps `config`) and as the
        # LAST key of the line (`summary`: what a reader of the line's tail sees first)
        summary, gates = flat_summary(out)
        out["config"].update({k: v for k, v in summary.items()})
        out["summary"] = summary
        print(json.dumps(out), flush=True)
        bad = [k for k, ok in gates.items() if ok is False]
        if bad:   # a fast path whose poses differ from the reference's is not a result
            print("bench.py: pose parity outside its bar in: %s" % ", ".join(bad), file=sys.stderr, flush=True)
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_live(args):
    """`--mode live`: one sequential-SLAM configuration as a line of its own (the default run carries both as blocks)."""
    sensor = "VLP-16" if args.sensor == SENSOR else args.sensor
    M = 200_000 if args.map_points == MAP_POINTS else args.map_points
    out = live_block(sensor, M, args.steps, args.warmup, cpu=not args.no_cpu_baseline, nodes=not args.no_live_nodes)
    out["config"]["env_overrides"] = env_overrides()
    from loam_velodyne_amd import loamx
    out["config"]["library_build"] = loamx.build_info()
    print(json.dumps(out), flush=True)
    pe = out.get("pose_err_vs_oracle")
    if pe and not pe.get("within_bar", True):
        print("bench.py: live-mode pose error outside its bar: %r" % pe, file=sys.stderr, flush=True)
        sys.exit(3)


def side_live(sensor, M, K, W, cpu=True, nodes=True):
    """A sequential-SLAM block of the default line, measured in a process of its own (`bench.py --mode live ...` as a child, its line
    embedded): inside the parent — after the batched windows have created and destroyed a few dozen HIP streams — the same chain ran
    1,230-1,353 sweeps/s (VLP-16) where a fresh process runs 1,440-1,510 (profiles/r06_ab.md section 8: the extraction + odometry wait is
    45 us longer, whatever the queue count or the NUMA binding).  LOAMX_BENCH_LIVE_INPROCESS=1 measures it inside the parent instead."""
    if os.environ.get("LOAMX_BENCH_LIVE_INPROCESS"):
        return live_block(sensor, M, K, W, cpu=cpu, nodes=nodes)
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "live", "--sensor", sensor, "--map-points", str(M), "--steps", str(K), "--warmup", str(W)]
    if not cpu:
        cmd.append("--no-cpu-baseline")
    if not nodes:
        cmd.append("--no-live-nodes")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}   # (the child is exactly `bench.py --mode live`: the runtime's default queues)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    if not lines:
        raise RuntimeError("live child produced no line (rc %d): %s" % (r.returncode, r.stderr[-300:]))
    blk = json.loads(lines[-1])
    blk["measured_in"] = "a child process: " + " ".join(cmd[1:])
    blk["child_rc"] = r.returncode   # (3: the child's own parity gate)
    return blk


def live_block(sensor, M, K, W, cpu=True, nodes=True):
    """BASELINE configs[1] / [2]: VLP-16 / 200 k-pt or HDL-32 / 500 k-pt LIVE map, ONE sweep in flight, sequential SLAM
    semantics — every sweep goes through the scan registration, the odometry and the mapping's process() (BasicLaserMapping.cpp:266-599:
    the map is updated and re-voxelised after every sweep, its grid index rebuilt for every sweep), the sweep from (pinned) host memory
    in and the registered cloud out to it, so the PCIe is inside the timed region.  CPU leg: the oracle's live-map process() on the same
    sweeps and the same initial map.  Returns the block (a bench line of its own under --mode live)."""
    import torch
    from loam_velodyne_amd import loamx, synth
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    world_model = synth.World(half_extent=65.0)
    cm, sm = world_model.make_map(M)
    T = 1 + W + K
    poses = synth.trajectory(T)
    jobs = [(65.0, sensor, poses[t], poses[t + 1], 500 + t) for t in range(T)]
    nw = max(1, min(int(os.environ.get("LOAMX_BENCH_WORKERS", "32")), len(os.sched_getaffinity(0)), T))
    if nw > 1:
        import multiprocessing as mp_
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=nw, mp_context=mp_.get_context("spawn"), initializer=unbind_worker) as ex:
            made = list(ex.map(synth.make_sweep_job, jobs, chunksize=max(1, T // (4 * nw))))
    else:
        made = [synth.make_sweep_job(j) for j in jobs]
    import types
    sweeps = [types.SimpleNamespace(points=p_, ring_sizes=r_) for p_, r_ in made]
    # the sweeps wait in, and the registered clouds land in, host memory the runtime has pinned (what a driver's receive buffers would be):
    # the library then copies straight from / to it instead of through a staging block of its own
    pts = [loamx.pinned_copy(sw.points) for sw in sweeps]
    landing = loamx.pinned_empty((max(len(sw.points) for sw in sweeps), 4))
    env_pool, env_future, env_paths = None, None, []

    def run_chain(linked):
        """one sweep in flight through the three handles; linked: the sweep's clouds go from handle to handle in HBM (loamx_*_process_linked)
        instead of through host arrays (the reference's ROS messages) — the sweep itself still comes from host memory and the registered
        full-resolution cloud still lands there"""
        import gc
        sr, od, mp = loamx.ScanRegistration(), loamx.LaserOdometry(), loamx.LaserMapping()
        mp.load_cubes(cm, sm)
        gc.collect()
        gc.disable()   # (as timeit does: a collection over this process's heap — torch is loaded — is a harness pause of tens of ms, not a property of the path)
        r = {"stage": np.zeros(3), "other": 0.0, "stats": [], "gn_ms": 0.0, "gn_launches": 0, "gn_qi": 0, "reg_ms": 0.0, "n_timed": 0, "poses": []}
        t0 = None
        for t in range(T):
            if t == 1 + W:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            sampled = t >= 1 + W and (t - (1 + W)) % TIMING_PERIOD == 0   # HIP-event pairs around the registration's launches on every TIMING_PERIOD-th sweep only (as in the batched mode)
            top = time.perf_counter()
            mp.set_timing(sampled)
            a = time.perf_counter()
            if linked:
                sr.process_linked(pts[t], sweeps[t].ring_sizes)
                b = time.perf_counter()      # (the extraction is only enqueued here: its time shows up in the odometry's share)
                od.process_linked(sr)
                c = time.perf_counter()
                mp.process_linked(od, landing)
            else:
                f = sr.process(pts[t], sweeps[t].ring_sizes)
                b = time.perf_counter()
                od.process(f)
                lc, ls = od.last_clouds()
                full = od.transform_to_end(f["full"])
                c = time.perf_counter()
                mp.update_odometry(od.transform_sum)
                mp.process(lc, ls, full, inplace=True)   # (full is transform_to_end's own array: registered where it lies, as the C entry point does)
            d = time.perf_counter()
            so_, sm_ = od.stats(), mp.stats()
            r["poses"].append((t, 0, np.array(od.transform_sum, np.float32), mp.transform("aft"), so_["iterations"], sm_["iterations"],
                               (so_["sel"], sm_["sel"], sm_["corner_ds"], sm_["surf_ds"])))   # (two 6-float reads and two small structs: ~3 us)
            if t >= 1 + W:
                r["stage"] += [b - a, c - b, d - c]
                r["stats"].append(mp.stats())
                if sampled:
                    tm = mp.timing()
                    r["gn_ms"] += tm["residual_ms"]; r["gn_launches"] += tm["residual_launches"]; r["gn_qi"] += tm["query_iterations"]; r["reg_ms"] += tm["run_ms"]
                    r["n_timed"] += 1
                r["other"] += (a - top) + (time.perf_counter() - d)
        torch.cuda.synchronize()
        r["elapsed"] = time.perf_counter() - t0
        gc.enable()
        r["aft"] = mp.transform("aft")
        r["speculation"] = mp.speculation()
        for h_ in (mp, od, sr):   # (free the handles' HIP streams before the next block makes its own)
            h_.close()
        return r

    host = run_chain(False)
    run = run_chain(True)
    assert np.array_equal(run["aft"], host["aft"]) and all(np.array_equal(p[2], q[2]) and np.array_equal(p[3], q[3]) and p[4:] == q[4:] for p, q in zip(run["poses"], host["poses"])), \
        "the linked chain and the host-message chain disagree"
    args = types.SimpleNamespace(no_live_nodes=not nodes, no_cpu_baseline=not cpu)
    stage, stats, gn_ms, gn_launches, gn_qi, reg_ms, n_timed = (run[k] for k in ("stage", "stats", "gn_ms", "gn_launches", "gn_qi", "reg_ms", "n_timed"))
    gpu_poses, elapsed = run["poses"], run["elapsed"]
    aft = run["aft"]
    out = {
        "metric": f"sweeps/sec (sequential SLAM: {sensor} sweep, {M // 1000}k-pt live map, one sweep in flight): feature extraction + odometry + mapping process()",
        "value": round(K / elapsed, 2), "unit": "sweeps/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{2 if sensor == 'HDL-32' else 1}]: {sensor} sweeps ({len(sweeps[0].points)} pts), {M}-pt LIVE map (updated, re-voxelised and re-indexed every sweep), "
                               "single-sweep entry points, the sweep from (pinned) host memory in and the registered full-resolution cloud out to it (PCIe inside "
                               "the timed region), the clouds between the three handles handed on in HBM (loamx_*_process_linked)",
                   "stage_ms_per_sweep": {"features_enqueue": round(stage[0] / K * 1e3, 4), "features_wait_and_odometry": round(stage[1] / K * 1e3, 4), "mapping": round(stage[2] / K * 1e3, 4),
                                          "harness_between_calls": round(run["other"] / K * 1e3, 4)},
                   "partition_prepared_ahead": {"adopted": run["speculation"][0], "redone": run["speculation"][1],
                                                "what": "sweeps (warm-up included) whose map partition + sub-map index had been built behind the previous sweep's "
                                                        "update for the predicted pose and were adopted because the true pose's plan was identical / sweeps that partitioned afresh"},
                   "host_message_chain": {"sweeps_per_s": round(K / host["elapsed"], 2), "ms_per_step": round(host["elapsed"] / K * 1e3, 4),
                                          "stage_ms_per_sweep": {"features": round(host["stage"][0] / K * 1e3, 4), "odometry": round(host["stage"][1] / K * 1e3, 4),
                                                                 "mapping": round(host["stage"][2] / K * 1e3, 4), "harness_between_calls": round(host["other"] / K * 1e3, 4)},
                                          "what": "the same sweeps with every cloud between the handles through host arrays, as the reference's nodes exchange ROS messages "
                                                  "(loamx_scanreg_process / loamx_odom_process / get_last_clouds / transform_to_end / loamx_map_process); poses, iteration "
                                                  "counts and the final map pose are asserted bit-identical to the linked chain's"},
                   "mean_map_iterations": round(float(np.mean([s["iterations"] for s in stats])), 2),
                   "mean_submap_points": round(float(np.mean([s["corner_from_map"] + s["surf_from_map"] for s in stats])), 1),
                   "registration_device_ms_per_sweep": round(reg_ms / max(n_timed, 1), 4),
                   "timing_sampling": f"HIP events around the registration and its Gauss-Newton launches on every {TIMING_PERIOD}th sweep ({n_timed} of {K})",
                   "final_pose_error_vs_ground_truth_m": round(float(np.abs(aft[3:] - poses[T, 3:]).max()), 4)},
    }
    # The same sweeps through the same three entry points, but run the way the reference runs them: three nodes, each consuming the
    # messages of the one before it as they arrive (scanRegistration of sweep t+2 || laserOdometry of t+1 || laserMapping of t; queues
    # of depth 2 between them).  Data flow and results are identical (asserted: the final pose equals the sequential run's bit for
    # bit); `value` above stays the one-sweep-in-flight figure.
    if not args.no_live_nodes:
        import queue
        import threading
        sr2, od2, mp2 = loamx.ScanRegistration(), loamx.LaserOdometry(), loamx.LaserMapping()
        mp2.load_cubes(cm, sm)
        q1, q2 = queue.Queue(maxsize=2), queue.Queue(maxsize=2)
        errs = []

        def node_scanreg():
            try:
                for t in range(T):
                    q1.put(sr2.process(sweeps[t].points, sweeps[t].ring_sizes))
            except BaseException as e:   # noqa: BLE001 (reported by the main thread)
                errs.append(e)
            q1.put(None)

        def node_odometry():
            try:
                while True:
                    f = q1.get()
                    if f is None:
                        break
                    od2.process(f)
                    lc2, ls2 = od2.last_clouds()
                    q2.put((lc2, ls2, od2.transform_to_end(f["full"]), od2.transform_sum))
            except BaseException as e:   # noqa: BLE001
                errs.append(e)
            q2.put(None)

        th = [threading.Thread(target=node_scanreg), threading.Thread(target=node_odometry)]
        for x in th:
            x.start()
        tn0 = None
        for t in range(T):
            item = q2.get()
            if item is None:
                break
            if t == 1 + W:
                tn0 = time.perf_counter()
            mp2.update_odometry(item[3])
            mp2.process(item[0], item[1], item[2], inplace=True)
        torch.cuda.synchronize()
        tn1 = time.perf_counter()
        for x in th:
            x.join()
        if errs:
            raise errs[0]
        aft2 = mp2.transform("aft")
        assert np.array_equal(aft, aft2), f"three concurrent nodes and the sequential run disagree: {aft} vs {aft2}"
        out["value_nodes_concurrent"] = round(K / (tn1 - tn0), 2)
        out["config"]["nodes_concurrent"] = ("the same entry points as three concurrent nodes (threads; queues of depth 2), as the reference's scanRegistration / "
                                             "laserOdometry / laserMapping run: throughput is set by the slowest node; final pose bit-identical to the sequential run")
    avg_launch_ms = gn_ms / max(gn_launches, 1)
    achieved = (72.0 * gn_qi / max(gn_launches, 1)) / (avg_launch_ms * 1e-3) / 1e9 if gn_launches else 0.0
    live_pmc, live_pmc_src = None, None
    for rnd in PROFILE_ROUNDS:   # PMC passes of THIS command (scripts/gpu_pmc.sh <tag> --mode live ...), committed per configuration, newest round first
        try:
            path_ = os.path.join(ROOT, "profiles", f"{rnd}_live_{sensor.lower().replace('-', '')}_pmc_summary.json")
            with open(path_) as f:
                live_pmc = json.load(f)["k_gn_iter_full_launch"]["traffic_bytes"]
            live_pmc_src = os.path.relpath(path_, ROOT)
            break
        except (OSError, KeyError, ValueError):
            continue
    out["roofline"] = {"kernel": "loamx::k_gn_iter", "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": live_pmc, "traffic_source": live_pmc_src, "peak_copy_ceiling": HBM_COPY_GBS,
                       "model": "72 B per query-iteration (12 B query + 5 x 12 B neighbours), S = 1 sweep per launch: one sweep's ~5 k queries "
                                "cannot fill the device — the launch is latency (search ~10 us + fit + 6x6 update step), not bandwidth",
                       "avg_launch_us": round(avg_launch_ms * 1e3, 3), "launches": gn_launches,
                       "algorithmic_bytes_per_launch": round(72.0 * gn_qi / max(gn_launches, 1), 1)}
    if not args.no_cpu_baseline:
        if all(len(sw.points) == len(sweeps[0].points) for sw in sweeps):   # the envelope chain: a process of its own beside the CPU legs below (after the timed device chains)
            import multiprocessing as mp_
            from concurrent.futures import ProcessPoolExecutor
            env_paths = [f"/dev/shm/loamx_bench_{os.getpid()}_live_{sensor}_sweeps.npy", f"/dev/shm/loamx_bench_{os.getpid()}_live_{sensor}_map.npy"]
            np.save(env_paths[0], np.stack([sw.points for sw in sweeps]))
            np.save(env_paths[1], np.concatenate([cm, sm], axis=0))
            env_pool = ProcessPoolExecutor(max_workers=1, mp_context=mp_.get_context("spawn"))
            env_future = env_pool.submit(chain_worker, ("oracle_fast", "live", env_paths[0], np.asarray(sweeps[0].ring_sizes), env_paths[1], len(cm), None, T))
        import oracle_py as op
        orc = op.Oracle(fast=True)
        osr, ood, omp = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc)
        omp.load_cubes(cm, sm)
        per = []
        for t in range(min(T, 1 + W + 20)):
            a = time.perf_counter()
            ood.set_features(osr.process(sweeps[t].points, sweeps[t].ring_sizes))
            ood.process()
            omp.set_inputs(ood.last_corner(), ood.last_surf(), ood.full_to_end(), ood.transform_sum)
            omp.process()
            if t >= 1 + W:
                per.append(time.perf_counter() - a)
        # ---- pose parity (BASELINE.json's metric, third part) in the live mode: the same sweeps, FREE RUNNING, through the oracle's parity
        # build (live map, every sweep inserted) — per sweep the difference of the mapped pose and of the accumulated odometry.  Free
        # running means a 1e-6-level difference can flip a point across a voxel face of the live map and the two chains then see slightly
        # different maps: the tests bound that drift at 2e-3 (tests/test_gpu_mapping.py) and check the per-step parity (both sides
        # started from the same state) at 1e-4; both figures of the run are reported here
        orc_p = op.Oracle(fast=False)
        psr, pod, pmp = op.ScanRegistration(orc_p), op.LaserOdometry(orc_p), op.LaserMapping(orc_p)
        pmp.load_cubes(cm, sm)
        chain = []
        for t in range(T):
            pod.set_features(psr.process(sweeps[t].points, sweeps[t].ring_sizes))
            pod.process()
            pmp.set_inputs(pod.last_corner(), pod.last_surf(), pod.full_to_end(), pod.transform_sum)
            pmp.process()
            so_, sm_ = pod.stats(), pmp.stats()
            chain.append((t, np.array(pod.transform_sum, np.float32), np.array(pmp.transform("aft"), np.float32), so_["iterations"], sm_["iterations"],
                          (so_["sel"], sm_["sel"], sm_["corner_ds"], sm_["surf_ds"])))
        pe = pose_error(gpu_poses, chain, stream=0)
        if pe:
            # the reference's own envelope over these sweeps in this mode: the oracle built as the reference's README builds the reference
            # (-O3 -march=native, FMA contraction) against its parity build, both free running over their own live maps
            env_rows = None
            try:
                _, env_rows = env_future.result(timeout=600) if env_future is not None else (None, None)
            except Exception as e:   # noqa: BLE001
                pe["reference_envelope"] = {"error": repr(e)[:160]}
            orc_rows = np.array([np.concatenate([[c[0]], c[1].astype(np.float64), c[2].astype(np.float64)]) for c in chain])
            env = reference_envelope({"oracle_fast": env_rows}, orc_rows) if env_rows is not None else None
            worst = max(pe["mapped_pose"]["max_m"], pe["mapped_pose"]["max_rad"])
            if env is not None:
                pe["reference_envelope"] = env
                pe["bar_free_running"] = max(1e-4, 2.0 * max(env["max_m"], env["max_rad"]))
                pe["bar_rule"] = ("free running over a LIVE map (the map itself depends on every earlier pose: a difference feeds back through the voxel grids and grows — "
                                  "the reference does this to itself): mapped pose <= max(1e-4, 2 x the difference the oracle shows between its -O3 -march=native build "
                                  "(the reference's README flags: FMA contraction) and its parity build over these very sweeps).  Per step from identical state (device and "
                                  "oracle given the same map and inputs) is the tests' job: tests/test_gpu_mapping.py, <= 1e-4 at VLP-16 / 200 k and HDL-32 / 500 k")
            else:
                pe["bar_free_running"] = 2e-3
                pe["bar_rule"] = "no envelope chain in this run: the tests' free-running bound, 2e-3 (tests/test_gpu_mapping.py)"
            pe["within_bar"] = bool(worst <= pe["bar_free_running"])
            pe.pop("sweeps_outside_bar_free_running", None)
            pe["note"] = ("free-running chains over %d sweeps with a LIVE map; " % len(chain)) + pe["note"]
        out["pose_err_vs_oracle"] = pe
        out["cpu_baseline"] = {"value": round(1.0 / float(np.median(per)), 4), "unit": "sweeps/s", "cores": 1, "kind": "port",
                               "sample": f"{len(per)} sweeps of the same sequence, same initial map, oracle live-map process() (g++ -O3 -march=native, one thread)",
                               "seconds_per_sweep": _stats(per), "host_cores_available": os.cpu_count()}
    if env_pool is not None:
        env_pool.shutdown(wait=False, cancel_futures=True)
        for p_ in env_paths:
            try:
                os.remove(p_)
            except OSError:
                pass
    return out


def bind_near_gpu(torch, local_rank):
    """Keep this process (and the pinned staging memory it first-touches) on the NUMA node of its GPU: on a two-socket host the
    driver's box showed the PCIe-inclusive window 30 % below the resident one when the staging buffers sat on the far socket.
    Returns the node, or None when the topology cannot be read."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        dom, bus, dev = getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", 0)
        if bus is None:
            return None
        with open(f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{dev:02x}.0/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        cpus = set()
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            for part in f.read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def copy_bandwidth(torch, dev, nbytes):
    """what the box's PCIe link gives a pinned <-> device copy of one step's size (GB/s each way, best of 5)"""
    h = torch.empty(nbytes // 4, dtype=torch.float32).pin_memory()
    d = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    out = {}
    for name, fn in (("h2d", lambda: d.copy_(h, non_blocking=True)), ("d2h", lambda: h.copy_(d, non_blocking=True))):
        best = 0.0
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); b.synchronize()
            best = max(best, nbytes / (a.elapsed_time(b) * 1e-3) / 1e9)
        out[name] = round(best, 1)
    return out


def pcie_inclusive_run(torch, loamx, local_rank, map_t, n_corner, n_surf, sweeps, starts, ns, W, K, dist, dev, lxdist):
    """K timed steps with host <-> device traffic inside the timed region: H2D of each step's sweeps (one pinned block per step, a
    copy stream, three steps ahead of the compute, handed over by a second host thread while the first one is inside step()) and
    D2H of each step's registered clouds (asynchronous, alternating device buffers, one pinned block)."""
    from concurrent.futures import ThreadPoolExecutor
    T = 1 + W + K
    AHEAD = 7                          # steps staged beyond the one being registered: the streaming ring's eight slots (look-ahead depth 6 + the one being staged)
    T_all = min(len(sweeps), T + AHEAD - 1)   # staged beyond the last step that runs: the look-ahead's work stays inside the window (main(): LOOK)
    n_pts = len(sweeps[0][0][0])
    assert all(len(sweeps[t][s][0]) == n_pts for t in range(T_all) for s in range(ns))
    pinned = []
    for t in range(T_all):   # the streams' clouds of a step lie back to back: the library hands the block over in one copy
        blk = torch.empty((ns * n_pts, 4), dtype=torch.float32).pin_memory()
        for s in range(ns):
            blk[s * n_pts:(s + 1) * n_pts].copy_(torch.from_numpy(np.ascontiguousarray(sweeps[t][s][0], np.float32)))
        views = blk.numpy()
        pinned.append(([(views[s * n_pts:(s + 1) * n_pts], sweeps[t][s][1]) for s in range(ns)], blk))
    out_blk = [torch.empty((ns * n_pts, 4), dtype=torch.float32).pin_memory() for _ in range(2)]
    outs = [[b.numpy()[s * n_pts:(s + 1) * n_pts] for s in range(ns)] for b in out_blk]
    p = loamx.Pipeline(ns, scanreg=dict(device=local_rank), odom=dict(device=local_rank), mapping=dict(device=local_rank))
    p.set_frozen_device(map_t.data_ptr(), n_corner, map_t.data_ptr() + 16 * n_corner, n_surf)
    for k in range(ns):
        p.set_state(k, aft=starts[k])
    p.enable_async_downloads()
    # the C-ABI arguments of every hand-over are built once, ahead of the loop: inside it the second thread only makes the library
    # call (which releases the GIL) — Python-level marshalling there would hold the GIL just when step() wants it back
    import ctypes as C
    L = loamx.lib()
    stage_args = []
    for t in range(T_all):
        pts = [loamx.as_points(a) for a, _ in pinned[t][0]]
        rings = [np.ascontiguousarray(r, np.uint32) for _, r in pinned[t][0]]
        CA = (loamx.Cloud * ns)(*[loamx.cloud_of(a) for a in pts])
        RP = (C.c_void_p * ns)(*[r.ctypes.data for r in rings])
        NR = (C.c_uint32 * ns)(*[len(r) for r in rings])
        stage_args.append((CA, RP, NR, pts, rings))
    out_args = [(loamx.Cloud * ns)(*[loamx.cloud_of(a) for a in outs[k]]) for k in range(2)]

    stage_s = []

    def stage(t):
        CA, RP, NR, _, _ = stage_args[t]
        ts0 = time.perf_counter()
        rc_ = L.loamx_pipeline_stage_step(p.h, t, CA, RP, NR)
        stage_s.append(time.perf_counter() - ts0)
        if rc_ < 0:
            raise RuntimeError(L.loamx_last_error().decode())

    for t in range(min(AHEAD, T_all)):
        stage(t)
    assert p.lookahead_depth() == AHEAD - 1, "the streaming ring's look-ahead depth changed: adapt AHEAD"
    stager = ThreadPoolExecutor(max_workers=1)   # stage_step(t + AHEAD) runs beside step(t)
    import gc
    t0 = None
    mapped_pts = 0
    host = np.zeros(3)
    gc.collect()
    gc.disable()   # (see the resident window)
    for t in range(T):
        if t == 1 + W:   # steady state: the pipeline is NOT emptied here (steps t .. t+6 are staged, the look-ahead has run as far as it
            p.drain_lookahead()   # may, copies may be in flight — as in production); the window ends the same way plus everything landed,
            if dist is not None:  # so the look-ahead work and the copies of exactly K steps are inside
                dist.barrier()
            t0 = time.perf_counter()
        ta = time.perf_counter()
        fut = stager.submit(stage, t + AHEAD) if t + AHEAD < T_all else None   # slot (t + AHEAD) % 8: free since step t - 1 has run
        rc = p.step(t)
        tb = time.perf_counter()
        if fut is not None:
            fut.result()
        tc = time.perf_counter()
        if rc == loamx.OK:
            CA = out_args[t & 1]
            for k in range(ns):
                CA[k].count = n_pts   # (capacity in, size out)
            if L.loamx_pipeline_download_step_async(p.h, CA, ns) < 0:
                raise RuntimeError(L.loamx_last_error().decode())
            if t >= 1 + W:
                mapped_pts += sum(int(CA[k].count) for k in range(ns))
        if t >= 1 + W:
            host += [tb - ta, tc - tb, time.perf_counter() - tc]
    p.drain_lookahead()
    p.wait_downloads()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    elapsed_local = elapsed
    elapsed = lxdist.max_over_ranks(elapsed, dist, dev)
    n_direct, n_hip = p.download_counts()
    world = dist.get_world_size() if dist is not None else 1
    bw = copy_bandwidth(torch, dev, int(ns * n_pts * 16))
    h2d_b, d2h_b = int(ns * n_pts * 16), int(mapped_pts * 16 // max(K, 1))
    p.close()
    return {
        "value": round(world * ns * K / elapsed, 2), "unit": "sweeps/s", "ms_per_step": round(elapsed / K * 1e3, 4),
        "h2d_bytes_per_step": h2d_b, "d2h_bytes_per_step": d2h_b,
        "link_gbps": bw,   # a pinned <-> device copy of one step's size on this box, each way
        "h2d_ms_per_step": round(h2d_b / (bw["h2d"] * 1e9) * 1e3, 4), "d2h_ms_per_step": round(d2h_b / (bw["d2h"] * 1e9) * 1e3, 4),
        "achieved_gbps_each_way": round(h2d_b / (elapsed / K) / 1e9, 2),
        "downloads": {"sdma_direct": n_direct, "hipMemcpyAsync": n_hip},
        "host_ms_per_step": {"inside_step": round(host[0] / K * 1e3, 4), "waiting_for_the_stager": round(host[1] / K * 1e3, 4), "download_call": round(host[2] / K * 1e3, 4),
                             "stage_call_on_the_stager_thread": round(float(np.mean(stage_s[AHEAD:])) * 1e3, 4) if len(stage_s) > AHEAD else None,
                             "window_close": round((elapsed_local - host.sum()) / K * 1e3, 4)},
        "note": "same workload and steps as `value`, but every step's sweeps cross PCIe inside the timed region (one pinned block per step, copy "
                "stream, staged seven steps ahead by a second host thread) and every step's registered full-resolution clouds are copied back "
                "(asynchronous, alternating buffers); steady-state window: not drained at its start, fully drained (downloads landed, device "
                "idle) at its end",
    }


CHAIN_KINDS = {   # CPU chains a reference envelope is made of: name -> (what runs, what it is compared with)
    "oracle_fast": "the oracle built the way the reference's README builds the reference (g++ -O3 -march=native: FMA contraction) against its parity build (-O2 -ffp-contract=off)",
    "ref": "the reference's own translation units (oracle/_ref/libref_{scanreg,odometry,mapping}.so) against the oracle's parity build — expected 0: the oracle is pinned bit for bit",
    "ref_map_alt": "the reference's units with the mapping's forwarded Eigen operations done the other plausible way (libref_mapping_alt.so: products accumulated in double, solve by elimination, Jacobi in double) against the plain units",
    "ref_odom_alt": "the same for the odometry (libref_odometry_alt.so) against the plain units",
    "ref_both_alt": "both alternative units against the plain units",
}


def chain_worker(job):
    """One CPU chain over one stream's sweeps, in a worker process (bench.py's CPU leg: oracle/ and oracle/_ref are the checker).
    job = (kind, mode, sweeps .npy path, ring sizes, corner map, surf map paths, start6 or None, T); mode "frozen": the batched mode's
    protocol (frozen sub-map, bench.py oracle_parity_chain), "live": the sequential-SLAM protocol (live map, every sweep inserted).
    -> rows [t, transformSum (6), mapped pose (6)] as float64 array (rows of the sweeps that were registered)."""
    kind, mode, sweeps_path, rings, map_path, n_corner, start, T = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    sw = np.load(sweeps_path, mmap_mode="r")
    m = np.load(map_path)
    rings = np.asarray(rings, np.uint32)
    if kind in ("oracle", "oracle_fast"):
        orc = op.Oracle(fast=(kind == "oracle_fast"))
        sr, od, mp = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc)
    else:
        if not (op.RefScanRegistration.available() and op.RefLaserOdometry.available() and op.RefLaserMapping.available()):
            return kind, None
        if kind in ("ref_odom_alt", "ref_both_alt") and not op.RefLaserOdometryAlt.available():
            return kind, None
        if kind in ("ref_map_alt", "ref_both_alt") and not op.RefLaserMappingAlt.available():
            return kind, None
        sr = op.RefScanRegistration()
        od = op.RefLaserOdometryAlt() if kind in ("ref_odom_alt", "ref_both_alt") else op.RefLaserOdometry()
        mp = op.RefLaserMappingAlt() if kind in ("ref_map_alt", "ref_both_alt") else op.RefLaserMapping()
    rows = []
    if mode == "frozen":
        mp.set_frozen(m[:n_corner], m[n_corner:])
        mp.set_transform("aft", np.asarray(start, np.float32))
    else:
        mp.load_cubes(m[:n_corner], m[n_corner:])
    for t in range(min(T, len(sw))):
        od.set_features(sr.process(np.asarray(sw[t]), rings))
        od.process()
        if mode == "frozen":
            if t > 0:
                mp.set_transform("sum", od.transform_sum)
                pose = mp.register_frozen(od.last_corner(), od.last_surf(), mp.associate())
                mp.set_transform("bef", od.transform_sum)
                mp.set_transform("aft", pose)
                rows.append(np.concatenate([[t], np.asarray(od.transform_sum, np.float64), np.asarray(pose, np.float64)]))
        else:
            mp.set_inputs(od.last_corner(), od.last_surf(), od.full_to_end(), od.transform_sum)
            mp.process()
            rows.append(np.concatenate([[t], np.asarray(od.transform_sum, np.float64), np.asarray(mp.transform("aft"), np.float64)]))
    return kind, np.array(rows)


class EnvelopeJobs:
    """The reference's own envelope over a chain of sweeps: the same sweeps through the reference's code in builds that differ only in what
    the reference does not pin (compiler contraction; the accumulation order inside the Eigen operations) — started as worker processes
    as soon as the sweeps exist, collected when the bench line is assembled (they run beside the other blocks; the box has the cores)."""

    def __init__(self, tag, mode, sweeps_arr, rings, cm, sm, start, T, kinds):
        import multiprocessing as mp_
        from concurrent.futures import ProcessPoolExecutor
        self.paths = [f"/dev/shm/loamx_bench_{os.getpid()}_{tag}_sweeps.npy", f"/dev/shm/loamx_bench_{os.getpid()}_{tag}_map.npy"]
        np.save(self.paths[0], sweeps_arr)
        np.save(self.paths[1], np.concatenate([cm, sm], axis=0))
        self.ex = ProcessPoolExecutor(max_workers=max(1, len(kinds)), mp_context=mp_.get_context("spawn"))
        self.t0 = time.perf_counter()
        self.futs = [self.ex.submit(chain_worker, (k, mode, self.paths[0], np.asarray(rings), self.paths[1], len(cm), start, T)) for k in kinds]

    def _signal(self, sig):
        import signal
        for pid in list(getattr(self.ex, "_processes", {}) or {}):
            try:
                os.kill(pid, getattr(signal, sig))
            except OSError:
                pass

    def pause(self):
        """stop the worker processes while a device window is timed (the batched pipeline's host threads are latency-sensitive)"""
        self._signal("SIGSTOP")

    def resume(self):
        self._signal("SIGCONT")

    def collect(self):
        self.resume()
        out = {}
        try:
            for f in self.futs:
                try:
                    k, rows = f.result(timeout=900)
                    if rows is not None:
                        out[k] = rows
                except Exception as e:   # an envelope chain that failed is missing from the block, never fatal to the line
                    out.setdefault("_errors", []).append(repr(e)[:160])
        finally:
            self.ex.shutdown(wait=False, cancel_futures=True)
            for p_ in self.paths:
                try:
                    os.remove(p_)
                except OSError:
                    pass
        self.seconds = time.perf_counter() - self.t0
        return out


def _pair_stats(X, Y):
    """rows [t, ts(6), pose(6)] of two chains -> statistics of the per-sweep max |difference| of the mapped pose and of the odometry"""
    ty = {int(r[0]): r for r in Y}
    P = np.array([[np.abs(r[10:13] - ty[int(r[0])][10:13]).max(), np.abs(r[7:10] - ty[int(r[0])][7:10]).max(),
                   np.abs(r[4:7] - ty[int(r[0])][4:7]).max(), np.abs(r[1:4] - ty[int(r[0])][1:4]).max()] for r in X if int(r[0]) in ty and int(r[0]) >= 1])
    if not len(P):
        return None
    tt = [int(r[0]) for r in X if int(r[0]) in ty and int(r[0]) >= 1]
    k = int(np.argmax(P[:, 0]))
    return {"sweeps": int(len(P)), "mapped_max_m": float(P[:, 0].max()), "mapped_max_rad": float(P[:, 1].max()), "mapped_rmse_m": float(np.sqrt((P[:, 0] ** 2).mean())),
            "mapped_p99_m": float(np.percentile(P[:, 0], 99)), "mapped_sweeps_above_1e-4": int((np.maximum(P[:, 0], P[:, 1]) > 1e-4).sum()), "mapped_max_at_sweep": tt[k],
            "odometry_sum_max_m": float(P[:, 2].max()), "odometry_sum_max_rad": float(P[:, 3].max())}


def reference_envelope(chains, orc_rows):
    """{kind: rows} of EnvelopeJobs.collect() + the oracle parity chain's rows -> the `reference_envelope` block: per pair the statistics of
    how far the reference's own poses move between two builds of its own code, and the envelope = the largest of them"""
    pairs = {}
    base_ref = chains.get("ref")
    for k, rows in chains.items():
        if k.startswith("_"):
            continue
        base = orc_rows if k in ("oracle_fast", "ref") or base_ref is None else base_ref
        st = _pair_stats(rows, base)
        if st:
            st["what"] = CHAIN_KINDS.get(k, k)
            pairs[k] = st
    env = [v for k, v in pairs.items() if k != "ref"]   # ("ref" against the oracle is the pin, not a sensitivity pair)
    return {"pairs": pairs,
            "max_m": max([v["mapped_max_m"] for v in env], default=0.0), "max_rad": max([v["mapped_max_rad"] for v in env], default=0.0),
            "rmse_m": max([v["mapped_rmse_m"] for v in env], default=0.0), "p99_m": max([v["mapped_p99_m"] for v in env], default=0.0),
            "sweeps_above_1e-4": max([v["mapped_sweeps_above_1e-4"] for v in env], default=0),
            "odometry_sum_max_m": max([v["odometry_sum_max_m"] for v in env], default=0.0),
            "errors": chains.get("_errors")}


def pose_error(gpu_poses, orc_poses, stream=0, envelope=None, per_step=None):
    """BASELINE.json metric part 3: the benchmarked path's poses against the oracle chain's (the CPU restatement of the reference, pinned
    bit for bit against the reference's own translation units by tests/test_ref_pinning.py — and, with an envelope, by the "ref" pair over
    this very chain) on the same sweeps of one stream from the same start: the mapped pose (transformAftMapped), the odometry's per-sweep
    step and the accumulated odometry (transformSum) after every sweep.
    Bar (north_star: 1e-4 m / 1e-4 rad "on identical sweeps"):
      * per step from identical state (per_step: the device registration run on the oracle chain's own inputs of every sweep): 1e-4, flat;
      * free running (both chains feed their own results forward, so a 1e-6 difference can flip a voxel face or a 5-NN set and the NEXT
        pose moves by more than the difference that caused it — the reference does this to itself between two builds of its own code):
        max(1e-4, the reference's envelope over the SAME sweeps) when an envelope was measured, 1e-4 otherwise."""
    g = {r[0]: r for r in gpu_poses if r[1] == stream}
    rows, steps, cnt_eq, cnts = [], [], [], []
    prev = None
    for r_o in orc_poses:
        t, ts_o, aft_o, oi_o, mi_o = r_o[:5]
        if t in g and t >= 1:
            r_g = g[t]
            ts_g, aft_g, oi_g, mi_g = r_g[2], r_g[3], r_g[4], r_g[5]
            inc = np.abs((ts_g - prev[0]) - (ts_o - prev[1])) if prev is not None else np.abs(ts_g - ts_o)   # (the chains start from the same transformSum)
            rows.append((np.abs(aft_g[3:] - aft_o[3:]).max(), np.abs(aft_g[:3] - aft_o[:3]).max(), np.abs(ts_g[3:] - ts_o[3:]).max(),
                         np.abs(ts_g[:3] - ts_o[:3]).max(), int(oi_g == oi_o), int(mi_g == mi_o), inc[3:].max(), inc[:3].max(),
                         int(np.argmax(np.abs(aft_g[3:] - aft_o[3:])))))
            steps.append(t)
            if len(r_g) > 6 and len(r_o) > 5:
                cnts.append((tuple(int(x) for x in r_g[6]), tuple(int(x) for x in r_o[5])))
                cnt_eq.append([int(x == y) for x, y in zip(r_g[6], r_o[5])])
            prev = (ts_g, ts_o)
        elif t in g:
            prev = (g[t][2], ts_o)
    if not rows:
        return None
    a = np.array(rows, float)
    d_sum = a[:, 2]
    inc = np.diff(np.concatenate([[0.0], d_sum]))
    k_big = int(np.argmax(inc))
    k_max = int(np.argmax(a[:, 0]))
    out = {"stream": stream, "sweeps": len(rows),
           "mapped_pose": {"max_m": float(a[:, 0].max()), "max_rad": float(a[:, 1].max()), "rmse_m": float(np.sqrt((a[:, 0] ** 2).mean())),
                           "rmse_rad": float(np.sqrt((a[:, 1] ** 2).mean())), "p99_m": float(np.percentile(a[:, 0], 99)),
                           "sweeps_above_1e-4": int((np.maximum(a[:, 0], a[:, 1]) > 1e-4).sum()),
                           "max_at": {"sweep": int(steps[k_max]), "component": "xyz"[int(a[k_max, 8])],
                                      "counts_gpu": (dict(zip(("odom_sel", "map_sel", "corner_ds", "surf_ds"), cnts[k_max][0])) if cnts else None),
                                      "counts_oracle": (dict(zip(("odom_sel", "map_sel", "corner_ds", "surf_ds"), cnts[k_max][1])) if cnts else None)}},
           "odometry_step": {"max_m": float(a[:, 6].max()), "max_rad": float(a[:, 7].max()),
                             "what": "per sweep: |(transformSum[t] - transformSum[t-1]) of the device - the same of the oracle| — the odometry's own output of that sweep"},
           "odometry_sum": {"max_m": float(a[:, 2].max()), "max_rad": float(a[:, 3].max()), "rmse_m": float(np.sqrt((a[:, 2] ** 2).mean())),
                            "rmse_rad": float(np.sqrt((a[:, 3] ** 2).mean())),
                            "largest_step": {"sweep": int(steps[k_big]), "increase_m": round(float(inc[k_big]), 7), "odometry_iterations_equal_there": bool(a[k_big, 4])},
                            "what": "the accumulated odometry integrates every sweep's step, so one sweep's 1e-5 stays in every later transformSum (a random walk over "
                                    "the chain, not a per-sweep error; the mapped pose does not inherit it: the registration re-anchors every sweep) — reported, "
                                    "gated only through odometry_step"},
           "odometry_iterations_equal": int(a[:, 4].sum()), "mapping_iterations_equal": int(a[:, 5].sum())}
    if cnt_eq:
        ce = np.array(cnt_eq).sum(0)
        out["counts_equal"] = dict(zip(("odom_sel", "map_sel", "corner_ds", "surf_ds"), (int(x) for x in ce)))
    # ---- the bar
    free = np.maximum.reduce([a[:, 0], a[:, 1], a[:, 6], a[:, 7]])   # per sweep: mapped pose (m, rad) and odometry step (m, rad)
    bar = 1e-4
    if envelope is not None:
        out["reference_envelope"] = envelope
        bar = max(1e-4, envelope["max_m"], envelope["max_rad"])
    out["bar_free_running"] = bar
    ps = per_step or {}
    by = ps.get("by_sweep") or {}
    outside = []
    ok = True
    for k in np.nonzero(free > bar)[0]:
        t = int(steps[k])
        d_ps = by.get(t)
        flip = bool(cnts and cnts[k][0] != cnts[k][1])
        explained = bool(envelope is not None and d_ps is not None and d_ps <= 1e-6 and flip)
        outside.append({"sweep": t, "difference": float(free[k]), "per_step_from_identical_state": d_ps, "a_discrete_count_differs": flip,
                        "counts_gpu": list(cnts[k][0]) if cnts else None, "counts_oracle": list(cnts[k][1]) if cnts else None, "explained_as_threshold_flip": explained})
        ok = ok and explained
    out["sweeps_outside_bar_free_running"] = outside
    if envelope is not None:   # the distribution must not be worse than what the reference shows against itself
        dist_ok = (out["mapped_pose"]["rmse_m"] <= max(envelope["rmse_m"], 1e-5) and out["mapped_pose"]["p99_m"] <= max(envelope["p99_m"], 1e-4)
                   and out["mapped_pose"]["sweeps_above_1e-4"] <= max(envelope["sweeps_above_1e-4"], 0))
        out["distribution_within_envelope"] = bool(dist_ok)
        ok = ok and dist_ok
    if per_step is not None:
        out["per_step_from_identical_state"] = {k_: v_ for k_, v_ in ps.items() if k_ != "by_sweep"}
        if ps.get("max_m") is not None:
            ok = ok and max(ps["max_m"], ps["max_rad"]) <= 1e-4
    out["bar"] = 1e-4
    out["within_bar"] = bool(ok)
    out["bar_rule"] = ("(1) per step from identical state (the device registration on the oracle chain's own inputs of EVERY sweep): <= 1e-4, flat.  (2) free running "
                       "(each chain feeds its own results forward): mapped pose and odometry step <= max(1e-4, reference_envelope.max) — the largest difference the "
                       "reference's own code shows against itself over these very sweeps between builds that differ only in what it does not pin (FMA contraction "
                       "under its README's -march=native; the accumulation order inside the Eigen operations); a sweep beyond that is accepted only as an explained "
                       "threshold flip: per step from identical state <= 1e-6 there AND a discrete count (selected rows / voxel-grid sizes) differs there; and the "
                       "distribution (rmse, p99, sweeps above 1e-4) must not exceed the envelope's.  Without an envelope (short window): 1e-4, flat."
                       if envelope is not None else "1e-4, flat, on the mapped pose and the odometry step (no envelope measured over this window)")
    out["note"] = ("per-sweep max |difference| of (x, y, z) and (rx, ry, rz) between the GPU pipeline (a separate, untimed window of the same steps) and the "
                   "oracle chain; bench.py exits with status 3 when any window is outside its bar")
    return out


def flat_summary(out):
    """({flat scalar figures of every block}, {block: within_bar or None}) of an assembled bench line"""
    sm, gates = {}, {}

    def pe_of(blk):
        return blk.get("pose_err_vs_oracle") if isinstance(blk, dict) and isinstance(blk.get("pose_err_vs_oracle"), dict) else None

    sm["value_sweeps_per_s"] = out.get("value"); sm["ms_per_step"] = out.get("ms_per_step"); sm["value_median_of_windows"] = out.get("value_median")
    if isinstance(out.get("pcie_inclusive"), dict):
        sm["pcie_inclusive_sweeps_per_s"] = out["pcie_inclusive"].get("value")
    if isinstance(out.get("roofline"), dict):
        sm["roofline_kernel"] = out["roofline"].get("kernel"); sm["roofline_frac"] = out["roofline"].get("frac")
    if isinstance(out.get("config"), dict):
        sm["path_hbm_frac"] = out["config"].get("path_hbm_frac")
    if isinstance(out.get("cpu_baseline"), dict):
        sm["cpu_sweeps_per_s_one_core"] = out["cpu_baseline"].get("value")
    pe = pe_of(out)
    if pe:
        sm["pose_max_m"] = pe["mapped_pose"]["max_m"]; sm["pose_max_rad"] = pe["mapped_pose"]["max_rad"]; sm["pose_within_bar"] = pe["within_bar"]
        gates["value (short window)"] = pe["within_bar"]
    vl = out.get("value_long")
    if isinstance(vl, dict):
        sm["value_long_sweeps_per_s"] = vl.get("value"); sm["value_long_ms_per_step"] = vl.get("ms_per_step"); sm["value_long_steps"] = vl.get("steps")
        pe = pe_of(vl)
        if pe:
            sm["value_long_pose_max_m"] = pe["mapped_pose"]["max_m"]; sm["value_long_pose_rmse_m"] = pe["mapped_pose"]["rmse_m"]
            sm["value_long_bar_free_running"] = pe.get("bar_free_running")
            if pe.get("reference_envelope"):
                sm["value_long_reference_envelope_max_m"] = pe["reference_envelope"]["max_m"]; sm["value_long_reference_envelope_rmse_m"] = pe["reference_envelope"]["rmse_m"]
                rp = pe["reference_envelope"]["pairs"].get("ref")
                if rp:
                    sm["value_long_reference_units_vs_oracle_max_m"] = rp["mapped_max_m"]
            if pe.get("per_step_from_identical_state"):
                sm["value_long_per_step_identical_state_max_m"] = pe["per_step_from_identical_state"]["max_m"]
            sm["value_long_sweeps_outside_bar"] = len(pe.get("sweeps_outside_bar_free_running") or [])
            sm["value_long_within_bar"] = pe["within_bar"]
            gates["value_long"] = pe["within_bar"]
    for key in ("map_2m", "live_vlp16", "live_hdl32"):
        blk = out.get(key)
        if not isinstance(blk, dict):
            continue
        if "error" in blk:
            sm[key + "_error"] = blk["error"][:120]
            continue
        sm[key + "_sweeps_per_s"] = blk.get("value"); sm[key + "_ms_per_sweep" if key.startswith("live") else key + "_ms_per_step"] = blk.get("ms_per_step")
        if isinstance(blk.get("cpu_baseline"), dict):
            sm[key + "_cpu_sweeps_per_s"] = blk["cpu_baseline"].get("value")
        if isinstance(blk.get("roofline"), dict):
            sm[key + "_roofline_frac"] = blk["roofline"].get("frac")
        pe = pe_of(blk)
        if pe:
            sm[key + "_pose_max_m"] = pe["mapped_pose"]["max_m"]; sm[key + "_bar"] = pe.get("bar_free_running"); sm[key + "_within_bar"] = pe["within_bar"]
            gates[key] = pe["within_bar"]
    sm["all_within_bar"] = all(v is not False for v in gates.values())
    return sm, gates


def env_overrides():
    """every LOAMX_* / GPU_MAX_HW_QUEUES / HSA_* variable set in this process: the library reads its tuning and tracing switches from the
    environment (DESIGN.md section 5 lists them; the result-changing diagnostics only exist in a LOAMX_DIAG build, see library_build)"""
    keep = {}
    for k, v in sorted(os.environ.items()):
        if k.startswith("LOAMX_") or k in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
            keep[k] = v
    return keep


def _committed(name):
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
        if os.path.exists(path):
            return path
    return None


def committed_kernel_stats():
    """{kernel name: (total ns, calls, average ns)} of the committed rocprofv3 --kernel-trace --stats run of this command"""
    import csv
    path = _committed("bench_kernel_stats.csv")
    out = {}
    if not path:
        return out, None
    try:
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Name"].split("(")[0].replace("void ", "").strip()
                out[name] = (float(r["TotalDurationNs"]), int(r["Calls"]), float(r["AverageNs"]))
    except (OSError, KeyError, ValueError):
        pass
    return out, os.path.relpath(path, ROOT)


def dominant_roofline(gn_block, ol, ns):
    """The bench line's `roofline`: the kernel with the largest TotalDurationNs in the committed kernel stats.  For the
    odometry's k_odom_lm — a latency chain, not a bandwidth kernel — the HBM fraction is reported as the contract asks AND a latency
    model beside it: microseconds per Gauss-Newton iteration measured live in this run against a stated floor."""
    stats, src = committed_kernel_stats()
    ours = {k: v for k, v in stats.items() if k.startswith("loamx::")}
    dom = max(ours.items(), key=lambda kv: kv[1][0])[0] if ours else "loamx::k_gn_iter"
    gn_block = dict(gn_block)
    gn_block["peak_copy_ceiling"] = HBM_COPY_GBS
    gn_block["frac_of_copy_ceiling"] = round(gn_block["achieved"] / HBM_COPY_GBS, 6)
    if dom.startswith("loamx::k_odom_corr_grid") and ol.get("corr_launches"):
        # the correspondence kernel leads the committed stats: its block from THIS run's HIP events, the iteration kernel's beside it
        n_l = max(ol["corr_launches"], 1)
        avg_us = ol["corr_ms"] / n_l * 1e3
        bytes_per_launch = ol["corr_features"] / n_l * (12 + 16 * 64)
        achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9
        return {"kernel": dom, "dominant_by": f"largest TotalDurationNs in {src} ({stats[dom][0] / 1e6:.2f} ms over {stats[dom][1]} launches of the profiled run)",
                "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                "peak_copy_ceiling": HBM_COPY_GBS, "frac_of_copy_ceiling": round(achieved / HBM_COPY_GBS, 6), "traffic": pmc_traffic("k_odom_corr_grid"),
                "traffic_note": f"bytes of one launch of {max(ns // 2, 1)} streams (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes of this command, {pmc_source()})",
                "model": "per feature 12 B query + ~64 candidate points x 16 B (the 27-cell block or the ring window), one wave per feature (SURVEY.md §8d): a chain of "
                         "~10 dependent memory round trips per wave, latency not bandwidth",
                "avg_launch_us": round(avg_us, 3), "launches": int(ol["corr_launches"]), "algorithmic_bytes_per_launch": round(bytes_per_launch, 1),
                "launch_sampling": f"HIP-event pairs around every launch of both odometry chains on every {TIMING_PERIOD}th step (launches over converged streams counted apart)",
                "noop_launches": {"n": int(ol["corr_noop_launches"]), "avg_us": round(ol["corr_noop_ms"] / max(ol["corr_noop_launches"], 1) * 1e3, 3)},
                "lm_pair": {"kernel": "loamx::k_odom_lm<1>", "avg_launch_us": round(ol["lm_ms"] / max(ol["lm_launches"], 1) * 1e3, 3), "launches": int(ol["lm_launches"]),
                            "us_per_iteration": round(ol["lm_ms"] * 1e3 / max(ol["lm_iterations"], 1), 3)},
                "k_gn_iter": gn_block}
    if not dom.startswith("loamx::k_odom_lm") or not ol.get("lm_launches"):
        gn_block["dominant_by"] = f"largest TotalDurationNs in {src}" if src else "no committed kernel stats found"
        return gn_block
    n_it, n_l = max(ol["lm_iterations"], 1), max(ol["lm_launches"], 1)
    avg_us = ol["lm_ms"] / n_l * 1e3
    bytes_per_launch = ol["lm_bytes"] / n_l
    achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9
    pm = pmc_traffic("k_odom_lm")
    return {
        "kernel": dom,
        "dominant_by": f"largest TotalDurationNs in {src} ({stats[dom][0] / 1e6:.2f} ms over {stats[dom][1]} launches of the profiled run)",
        "bound": "hbm",
        "achieved": round(achieved, 3),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 6),
        "peak_copy_ceiling": HBM_COPY_GBS,
        "frac_of_copy_ceiling": round(achieved / HBM_COPY_GBS, 6),
        "traffic": pm,
        "traffic_note": f"bytes of one launch of {max(ns // 2, 1)} streams (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes of this command, {pmc_source()}): "
                        "several times the algorithmic bytes — the workgroups of a stream poll each other's tagged records",
        "model": "48 B per feature (12 B query + 3 x 12 B correspondents, SURVEY.md §8d) read ONCE per launch of up to 5 iterations by the streams "
                 "still iterating; the features then stay in registers — the kernel is a chain of dependent iterations, not a stream of bytes",
        "avg_launch_us": round(avg_us, 3),
        "launches": int(ol["lm_launches"]),
        "launch_sampling": f"HIP-event pairs around every k_odom_lm launch of both odometry chains on every {TIMING_PERIOD}th step (launches over "
                           "converged streams counted apart)",
        "algorithmic_bytes_per_launch": round(bytes_per_launch, 1),
        "noop_launches": {"n": int(ol["lm_noop_launches"]), "avg_us": round(ol["lm_noop_ms"] / max(ol["lm_noop_launches"], 1) * 1e3, 3)},
        "latency_model": {
            "us_per_iteration": round(ol["lm_ms"] * 1e3 / n_it, 3),
            "iterations": int(ol["lm_iterations"]),
            "floor_us": 4.5,
            "floor_terms_us": {"residual rows (one feature per thread, ~350 dependent VALU instructions)": 0.6,
                               "28 sums over 256 rows through LDS (two barriers, 32 + 8 dependent double adds)": 0.5,
                               "exchange between the stream's 9 workgroups (one agent-scope store -> load round trip across XCDs)": 1.0,
                               "6x6 column-pivoted QR on one wave (~1,100 dependent instructions, bit-identical to the scalar routine)": 2.0,
                               "update, stop test, next sin/cos (lanes 0-5 of the same wave)": 0.4},
            "measured_terms_us": "profiles/r05_lm_stamps.md (in-kernel time stamps, -DLOAMX_PROF_LM; the iteration is unchanged since): rows 1.5, sums 1.0, record stores 0.4, poll 1.0-2.0, "
                                 "normal equations 0.6, QR 3.0, update 1.1",
            "source": "us_per_iteration: live, this run (sum of the timed launches' durations / iterations of each launch's slowest stream, "
                      "launch start-up included); floor: instruction counts of the kernel's serial chain at ~4.5 cycles per dependent wave64 "
                      "instruction and 2.4 GHz, plus the measured cross-XCD round trip (scripts/micro/atomics.hip)",
        },
        "corr_pair": {"kernel": "loamx::k_odom_corr_grid", "avg_launch_us": round(ol["corr_ms"] / max(ol["corr_launches"], 1) * 1e3, 3),
                      "launches": int(ol["corr_launches"]), "features_per_launch": round(ol["corr_features"] / max(ol["corr_launches"], 1), 1),
                      "noop_avg_us": round(ol["corr_noop_ms"] / max(ol["corr_noop_launches"], 1) * 1e3, 3)},
        "k_gn_iter": gn_block,
    }


def roofline_kernels(main, ns):
    """The other kernels on the critical chains next to the dominant one: algorithmic bytes per launch (SURVEY.md §8d models, stated per
    kernel), average launch duration and HBM-side traffic from the COMMITTED rocprofv3 passes of this command (bench.py cannot profile
    itself: profiles/<round>_bench_kernel_stats.csv, profiles/<round>_pmc_summary.json, newest round first; None when a file is missing)."""
    stats, src = committed_kernel_stats()
    dur = {k: v[2] / 1e3 for k, v in stats.items()}
    pmc, pmc_src = {}, None
    try:
        pmc_src = _committed("pmc_summary.json")
        with open(pmc_src) as f:
            pmc = json.load(f)
    except (OSError, ValueError, TypeError):
        pass
    feats = 36 * 64
    models = [   # kernel, algorithmic bytes per launch of `ns` sweeps, model
        ("loamx::k_odom_corr_grid", ns // 2 * feats * (12 + 16 * 64), "per feature: 12 B query + ~64 candidate points x 16 B (27-cell block or ring window), 4 streams per launch"),
        ("loamx::k_odom_lm<1>", ns // 2 * feats * 48, "per feature 48 B (query + tripod) once per launch of up to 5 iterations, 4 streams per launch"),
        ("loamx::k_vb_reduce", ns * 35000 * (16 + 8), "per stack point 16 B read + ~8 B of voxel means written"),
        ("loamx::k_feat_ring", ns * 131072 * 8, "per sweep point ~8 B (curvature + flags)"),
    ]
    out = []
    for name, nbytes, model in models:
        if main and str(main.get("kernel", "")).startswith(name):
            continue   # (one fraction per kernel per line: the dominant kernel's is `roofline`'s, measured live)
        us = next((v for k, v in dur.items() if k.startswith(name)), None)
        tr = next((v.get("traffic_bytes") for k, v in pmc.items() if isinstance(v, dict) and name.split("::")[-1].split("<")[0] in k), None)
        out.append({"kernel": name, "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_us": us,
                    "achieved_gbs": (round(nbytes / (us * 1e-6) / 1e9, 2) if us else None),
                    "frac": (round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 6) if us else None), "traffic": tr, "model": model,
                    "source": f"{src} (rocprofv3 --kernel-trace --stats of this command), {os.path.relpath(pmc_src, ROOT) if pmc_src else None}"})
    return out


def pmc_source():
    """the committed PMC summary pmc_traffic() reads (newest round that has one)"""
    for rnd in PROFILE_ROUNDS + ("r03",):
        if os.path.exists(os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.json")):
            return f"profiles/{rnd}_pmc_summary.json"
    return None


def pmc_traffic(kernel="k_gn_iter"):
    """HBM bytes per full launch of a kernel from the committed PMC passes of this same command (profiles/<round>_pmc_summary.json:
    FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc runs, FETCH_SIZE doubled per MI355X_MICROARCH.md).  bench.py cannot
    run the profiler on itself, so the figure is read, not measured live; None when no file has it."""
    for rnd in PROFILE_ROUNDS + ("r03",):
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary.json")) as f:
                d = json.load(f)
            if kernel == "k_gn_iter":
                return d["k_gn_iter_full_launch"]["traffic_bytes"]
            for k, v in d.items():
                if isinstance(v, dict) and kernel in k and v.get("traffic_bytes") is not None:
                    return v["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def _stats(x):
    x = np.asarray(x, float)
    return {"median": round(float(np.median(x)), 5), "p95": round(float(np.percentile(x, 95)), 5), "mean": round(float(x.mean()), 5), "n": int(len(x))}


def oracle_parity_chain(sweeps, starts, m, n_corner, T, stream=0, inputs=None):
    """the parity chain: the SAME sweeps of one stream through the oracle's parity build (liboracle.so: -O2 -ffp-contract=off, the build that
    is pinned bit for bit against the reference's translation units; the timed build of cpu_baseline is -O3 -march=native and contracts
    FMAs) -> [(step, transformSum, transformAftMapped, odometry iterations, mapping iterations, (odometry rows, mapping rows, corner / surface
    voxel-grid sizes))]; inputs (a list): receives (step, corner_last, surf_last, guess) of every registration — what the per-step check
    hands to the device"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    orc_p = op.Oracle(fast=False)
    psr, pod, pmp = op.ScanRegistration(orc_p), op.LaserOdometry(orc_p), op.LaserMapping(orc_p)
    pmp.set_frozen(m[:n_corner], m[n_corner:])
    pmp.set_transform("aft", starts[stream])
    out = []
    for t in range(T):
        pod.set_features(psr.process(*sweeps[t][stream]))
        pod.process()
        if t > 0:
            pmp.set_transform("sum", pod.transform_sum)
            lc, ls, guess = pod.last_corner(), pod.last_surf(), pmp.associate()
            pose = pmp.register_frozen(lc, ls, guess)
            pmp.set_transform("bef", pod.transform_sum)
            pmp.set_transform("aft", pose)
            so_, sm_ = pod.stats(), pmp.stats()
            out.append((t, np.array(pod.transform_sum, np.float32), np.array(pose, np.float32), so_["iterations"], sm_["iterations"],
                        (so_["sel"], sm_["sel"], sm_["corner_ds"], sm_["surf_ds"])))
            if inputs is not None:
                inputs.append((t, lc.copy(), ls.copy(), np.array(guess, np.float32)))
    return out


def per_step_check(loamx, cm, sm, inputs, orc_chain, chunk=32):
    """Per step from identical state: the device's batched registration (loamx_batch_*: the same kernels the pipeline runs) on the oracle
    chain's OWN inputs of every sweep — its re-projected clouds and its guess — against the oracle's pose of that sweep.  No feedback:
    whatever differs here is the registration's arithmetic on identical inputs."""
    want = {r[0]: r[2] for r in orc_chain}
    by = {}
    b = loamx.Batch(chunk)
    try:
        b.set_frozen(cm, sm)
        for a0 in range(0, len(inputs), chunk):
            part = inputs[a0:a0 + chunk]
            b.upload([x[1] for x in part], [x[2] for x in part], np.array([x[3] for x in part], np.float32))
            b.run()
            gp, _ = b.download()
            for k, x in enumerate(part):
                d = np.abs(np.asarray(gp[k], np.float64) - want[x[0]].astype(np.float64))
                by[int(x[0])] = (float(d[3:].max()), float(d[:3].max()))
    finally:
        b.close()
    dm = np.array([v[0] for v in by.values()])
    dr = np.array([v[1] for v in by.values()])
    t_max = max(by, key=lambda t_: by[t_][0])
    return {"sweeps": len(by), "max_m": float(dm.max()), "max_rad": float(dr.max()), "rmse_m": float(np.sqrt((dm ** 2).mean())), "max_at_sweep": int(t_max),
            "sweeps_above_1e-6": int((dm > 1e-6).sum()), "sweeps_above_1e-5": int((dm > 1e-5).sum()),
            "what": "loamx_batch_* (the pipeline's registration kernels) on the oracle chain's own (corner_last, surf_last, guess) of every sweep vs the oracle's pose",
            "by_sweep": {t_: max(v) for t_, v in by.items()}}


def cpu_baseline(sweeps, starts, map_t, n_measure=20, n_warm=3, n_reference=6, poses_out=None):
    """SURVEY.md §8(d) "CPU baseline timing": the oracle (CPU restatement of the reference, g++ -O3 -march=native, every stage
    single-threaded like the reference's nodes) on stream 0 of the SAME workload — 3 warm-up sweeps, then up to 20 measured
    ones against the same frozen map; per-stage median and p95; sweeps/s as the serial sum (1 core) and as the slowest stage
    (the reference's 3-process pipeline, 3 cores busy).  The kd-tree build over the map is timed separately (the GPU side also
    indexes the map once per epoch, outside the timed steps).  Beside it, where oracle/_ref was shipped, the reference's OWN
    translation units (BasicScanRegistration / BasicLaserOdometry / BasicLaserMapping.cpp compiled where they lie over header
    stand-ins, oracle/Makefile) on the first sweeps of the same stream: reported, not used as `value` — the stand-in containers
    make them slower than a real PCL / Eigen build would be, so the faster oracle is the conservative baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    orc = op.Oracle(fast=True)
    m = map_t.cpu().numpy()
    n_corner = int(round(len(m) * 0.1))
    osr, ood, omp = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc)
    t0 = time.perf_counter()
    omp.set_frozen(m[:n_corner], m[n_corner:])
    t_build = time.perf_counter() - t0
    omp.set_transform("aft", starts[0])
    T = min(len(sweeps), 1 + n_warm + n_measure)
    st = {"features": [], "odometry": [], "registration": []}
    for t in range(T):
        a = time.perf_counter()
        f = osr.process(*sweeps[t][0])
        b = time.perf_counter()
        ood.set_features(f)
        ood.process()
        c = time.perf_counter()
        if t > 0:
            omp.set_transform("sum", ood.transform_sum)
            pose = omp.register_frozen(ood.last_corner(), ood.last_surf(), omp.associate())
            omp.set_transform("bef", ood.transform_sum)
            omp.set_transform("aft", pose)
        d = time.perf_counter()
        if t > n_warm:
            st["features"].append(b - a); st["odometry"].append(c - b); st["registration"].append(d - c)
    if poses_out is not None:
        poses_out.extend(oracle_parity_chain(sweeps, starts, m, n_corner, T))
    per = np.array(st["features"]) + np.array(st["odometry"]) + np.array(st["registration"])
    med = {k: float(np.median(v)) for k, v in st.items()}
    serial = 1.0 / float(np.median(per))
    pipelined = 1.0 / max(med.values())
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {
        "value": round(serial, 4),
        "unit": "sweeps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"stream 0: {len(per)} measured sweeps of the benchmarked sensor after {n_warm} warm-up sweeps (+1 initialising), same frozen map; "
                  "kd-tree build excluded; value = 1 / median(features + odometry + registration) on one core",
        "pipelined_value": round(pipelined, 4),
        "pipelined_cores": 3,
        "pipelined_note": "1 / slowest stage median: the reference runs the three stages as three single-threaded ROS nodes",
        "seconds_per_sweep": _stats(per),
        "stage_seconds": {k: _stats(v) for k, v in st.items()},
        "kdtree_build_seconds": round(t_build, 4),
        "host_cores_available": os.cpu_count(),
        "cpu_model": cpu_model,
        "compiler_flags": "g++ -O3 -march=native (oracle/liboracle_fast.so)",
    }
    # ---- the reference's own translation units, timed beside the oracle
    try:
        if n_reference > 0 and op.RefScanRegistration.available() and op.RefLaserOdometry.available() and op.RefLaserMapping.available():
            rsr, rod, rmp = op.RefScanRegistration(), op.RefLaserOdometry(), op.RefLaserMapping()
            rmp.set_frozen(m[:n_corner], m[n_corner:])
            rmp.set_transform("aft", starts[0])
            rs = {"features": [], "odometry": [], "registration": []}
            for t in range(min(len(sweeps), 2 + n_reference)):
                a = time.perf_counter()
                f = rsr.process(*sweeps[t][0])
                b = time.perf_counter()
                rod.set_features(f)
                rod.process()
                c = time.perf_counter()
                if t > 0:
                    rmp.set_transform("sum", rod.transform_sum)
                    pose = rmp.register_frozen(rod.last_corner(), rod.last_surf(), rmp.associate())   # (rebuilds its kd-trees per call, BasicLaserMapping.cpp:636-637)
                    rmp.set_transform("bef", rod.transform_sum)
                    rmp.set_transform("aft", pose)
                d = time.perf_counter()
                if t > 1:
                    rs["features"].append(b - a); rs["odometry"].append(c - b); rs["registration"].append(d - c)
            rper = np.array(rs["features"]) + np.array(rs["odometry"]) + np.array(rs["registration"])
            out["reference_units"] = {
                "kind": "reference",
                "value": round(1.0 / float(np.median(rper)), 4),
                "unit": "sweeps/s",
                "sample": f"{len(rper)} sweeps of stream 0; registration includes the reference's per-call kd-tree rebuild over the map",
                "stage_seconds": {k: _stats(v) for k, v in rs.items()},
                "note": "oracle/_ref/libref_{scanreg,odometry,mapping}.so = the reference's sources over header stand-ins (slower than real PCL / Eigen)",
            }
    except Exception as e:   # the baseline of record is the oracle above; a failure here must not lose the bench line
        out["reference_units"] = {"error": repr(e)[:200]}
    return out


if __name__ == "__main__":
    main()
