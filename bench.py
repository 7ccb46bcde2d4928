#!/usr/bin/env python
"""bench.py — sweeps/sec of the registration hot path (feature extraction + odometry + scan-to-map registration).

Workload at every N (weak scaling, BASELINE.json configs[3] = "HDL-64 sweep, 1M-pt map, batch 32 over 4 GPUs", i.e.
8 sweeps in flight per GPU): each GPU runs 8 independent streams of synthetic HDL-64E sweeps (64 x 2048 = 131,072
points) against a frozen 1,000,000-point sub-map.  A "step" advances every stream by one sweep through the whole path:
  features (BasicScanRegistration) -> odometry (BasicLaserOdometry) -> registration (BasicLaserMapping, frozen map).
All sweeps are staged in HBM before the timed region; the map is generated on rank 0 and shipped to the other ranks
with one RCCL broadcast (torch.distributed, backend "nccl" = RCCL over xGMI) — the only collective on the path; the
streams themselves are sharded with no data-path exchange.

One JSON line on rank 0 (the driver's contract) plus `roofline` (dominant kernel: k_knn5, algorithmic bytes =
72 B x query-iterations, duration from HIP events on the library's own stream) and `cpu_baseline` (the oracle = CPU
restatement of the reference, -O3 -march=native, single thread, on a bounded sample of the same workload).
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
HANDLES_PER_GPU = 1      # >1: split the streams over several pipeline handles driven from host threads (measured: no gain)
SENSOR = "HDL-64E"
MAP_POINTS = 1_000_000
HBM_PEAK_GBS = 8000.0


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
    ap.add_argument("--sensor", default=SENSOR, choices=["HDL-32", "HDL-64E", "VLP-16"], help="parity / side configurations (BASELINE configs[1], [2])")
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU)
    ap.add_argument("--handles", type=int, default=HANDLES_PER_GPU,
                    help="pipeline handles per GPU, each on its own HIP stream and host thread, sharing the streams evenly")
    args = ap.parse_args()

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

    from loam_velodyne_amd import loamx, synth
    from loam_velodyne_amd import dist as lxdist

    ns, K, W = args.streams, args.steps, args.warmup
    T = 1 + W + K   # first sweep of a stream only initialises the odometry
    world_model = synth.World(half_extent=125.0)

    # ---- frozen map: generated on rank 0, broadcast over RCCL, adopted in place by the library
    M = args.map_points
    n_corner, n_surf = lxdist.split_map(M)
    map_t = torch.empty((M, 4), dtype=torch.float32, device=dev)
    if rank == 0:
        cm, sm = world_model.make_map(M)
        assert len(cm) == n_corner and len(sm) == n_surf
        map_t.copy_(torch.from_numpy(np.concatenate([cm, sm], axis=0)))
    t_bcast = 0.0
    if dist is not None:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lxdist.broadcast_map(map_t, dist, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0

    # ---- this rank's streams and staged sweeps (distinct trajectories per rank and stream)
    sweeps = [[None] * ns for _ in range(T)]
    starts = []
    for s, gs in enumerate(lxdist.stream_ids(rank, world, ns)):
        start = lxdist.stream_start(gs)
        poses = synth.trajectory(T, start=start)
        starts.append(np.array([0, 0, 0, start[0], start[1], start[2]], np.float32))
        for t in range(T):
            sw = synth.make_sweep(world_model, args.sensor, poses[t], poses[t + 1], seed=1000 * gs + t)
            sweeps[t][s] = (sw.points, sw.ring_sizes)
    n_points = len(sweeps[0][0][0])

    H = max(1, min(args.handles, ns))
    assert ns % H == 0, "--streams must be a multiple of --handles"
    per = ns // H
    torch.cuda.synchronize()
    pipes = []
    for h in range(H):
        p = loamx.Pipeline(per, scanreg=dict(device=local_rank), odom=dict(device=local_rank), mapping=dict(device=local_rank))
        p.set_frozen_device(map_t.data_ptr(), n_corner, map_t.data_ptr() + 16 * n_corner, n_surf)
        for k in range(per):
            p.set_state(k, aft=starts[h * per + k])
        p.upload([[sweeps[t][h * per + k] for k in range(per)] for t in range(T)])
        p.set_timing(True)
        pipes.append(p)

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=H)

    def run_step(t):
        if H == 1:
            pipes[0].step(t)
        else:
            list(pool.map(lambda p: p.step(t), pipes))   # ctypes releases the GIL: the handles really run concurrently

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (includes every stream's initialising first sweep)
    for t in range(1 + W):
        run_step(t)
    sync_all()
    stage = np.zeros(4)
    res_ms = 0.0
    res_launches = 0
    q_iters = 0
    queries = 0
    # double-buffered map epochs (off by default): epoch k+1 is broadcast and indexed in the background during epoch k
    # and swapped in before the first step of epoch k+1
    E = args.map_epoch_steps
    map_nexts = [torch.empty_like(map_t), torch.empty_like(map_t)] if E > 0 else None   # ping-pong: a buffer is rewritten only
    # after a registration against the index built from it has been observed complete
    ev_map = torch.cuda.Event() if E > 0 else None
    if E > 0:   # one untimed stage + swap so that the second set of index buffers exists before the timed region
        for p in pipes:
            p.stage_frozen_device(map_t.data_ptr(), n_corner, map_t.data_ptr() + 16 * n_corner, n_surf)
        torch.cuda.synchronize()
        for p in pipes:
            p.swap_frozen()
        sync_all()
    n_epochs = 0
    t0 = time.perf_counter()
    for t in range(1 + W, T):
        if E > 0:
            k = (t - (1 + W)) % E
            if k == 0:
                for p in pipes:
                    if p.swap_frozen():
                        n_epochs += 1
                map_next = map_nexts[((t - (1 + W)) // E) % 2]
                if rank == 0:
                    map_next.copy_(map_t, non_blocking=True)   # (the next epoch's map: same content, new buffer)
                if dist is not None:
                    dist.broadcast(map_next, src=0, async_op=True).wait()   # orders torch's stream behind RCCL's, not the host
                ev_map.record()
                for p in pipes:   # the index build waits for the event on the device; nothing blocks here
                    p.stage_frozen_device(map_next.data_ptr(), n_corner, map_next.data_ptr() + 16 * n_corner, n_surf, ev_map.cuda_event)
        run_step(t)
        for p in pipes:   # event read-back of the step that just finished (the step itself is synchronous)
            tm = p.timing()
            stage += np.array([tm["features_ms"], tm["odometry_ms"], tm["registration_ms"], tm["step_ms"]]) / H
            res_ms += tm["residual_ms"]
            res_launches += tm["residual_launches"]
            q_iters += tm["query_iterations"]
            queries += tm["queries"]
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed = lxdist.max_over_ranks(elapsed, dist, dev)

    # pose sanity of this rank's streams against ground truth (not the parity check — that is tests/)
    stats = [p.get(k)[3] for p in pipes for k in range(per)]
    sweeps_total = world * ns * K
    value = sweeps_total / elapsed

    if rank == 0:
        iters_map = np.mean([st["map_iterations"] for st in stats])
        iters_odom = np.mean([st["odom_iterations"] for st in stats])
        q_per_sweep = queries / max(ns * K, 1)
        # BASELINE.md / SURVEY.md §8d algorithmic bytes per registered sweep (S = streams sharing the frozen map epoch)
        k_feat = 36 * synth.SENSORS[args.sensor][0]   # (2 sharp + 4 flat) x 6 regions per ring
        bytes_per_sweep = 32 * n_points + 16 * M / (ns * K) + 72 * (q_iters / max(ns * K, 1)) + 48 * iters_odom * k_feat
        avg_launch_ms = res_ms / max(res_launches, 1)
        achieved = (72.0 * q_iters / max(res_launches, 1)) / (avg_launch_ms * 1e-3) / 1e9 if res_launches else 0.0
        out = {
            "metric": "sweeps/sec (64-ring, 1M-pt map): feature extraction + odometry + scan-to-map registration",
            "value": round(value, 2),
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.sensor} {synth.SENSORS[args.sensor][0]}x{synth.SENSORS[args.sensor][1]} sweeps ({n_points} pts), {M}-pt frozen sub-map, {ns} streams/GPU "
                            "(BASELINE configs[3]: batch 32 over 4 GPUs), full path per sweep",
                "streams_per_gpu": ns,
                "handles_per_gpu": H,
                "sweep_points": n_points,
                "map_points": M,
                "mean_map_iterations": round(float(iters_map), 2),
                "mean_odom_iterations": round(float(iters_odom), 2),
                "mean_queries_per_sweep": round(float(q_per_sweep), 1),
                "stage_ms_per_step": {"features": round(stage[0] / K, 4), "odometry": round(stage[1] / K, 4),
                                      "registration": round(stage[2] / K, 4), "gpu_step": round(stage[3] / K, 4)},
                "map_broadcast_ms": round(t_bcast * 1e3, 3),
                "map_epoch_steps": E,
                "map_epochs_swapped": n_epochs,
                "path_algorithmic_bytes_per_sweep": round(float(bytes_per_sweep), 1),
                "path_hbm_frac": round(float(bytes_per_sweep * value / world / (HBM_PEAK_GBS * 1e9)), 6),
            },
            "roofline": {
                "kernel": "loamx::k_knn5",
                "bound": "hbm",
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": pmc_traffic(),
                "traffic_note": "bytes of one FULL-SEARCH launch (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, "
                                "profiles/r01_pmc_summary.json); achieved/avg_launch_us average over all timed launches incl. "
                                "the no-op launches after convergence",
                "avg_launch_us": round(avg_launch_ms * 1e3, 3),
                "launches": res_launches,
                "algorithmic_bytes_per_launch": round(72.0 * q_iters / max(res_launches, 1), 1),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sweeps, starts, map_t)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic():
    """HBM bytes per full-search launch of the dominant kernel from the committed PMC passes of this same command
    (profiles/r01_pmc_summary.json: FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc runs, FETCH_SIZE
    doubled per MI355X_MICROARCH.md).  bench.py cannot run the profiler on itself, so the figure is read, not measured
    live; None when the file is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
            return json.load(f)["k_knn5_full_search_launch"]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(sweeps, starts, map_t):
    """The oracle (CPU restatement of the reference, -O3 -march=native, one thread) on a bounded sample of the same
    workload: stream 0, its first 3 sweeps (1 initialising + 2 registered) against the same 1M-pt map; the kd-tree build
    over the map is timed separately (the GPU side also builds its map index outside the timed steps)."""
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
    per = []
    stage = np.zeros(3)
    for t in range(3):
        a = time.perf_counter()
        f = osr.process(*sweeps[t][0])
        b = time.perf_counter()
        ood.set_features(f)
        ood.process()
        c = time.perf_counter()
        if t > 0:
            omp.set_transform("sum", ood.transform_sum)
            omp.register_frozen(ood.last_corner(), ood.last_surf(), omp.associate())
        d = time.perf_counter()
        if t > 0:
            per.append(d - a)
            stage += [b - a, c - b, d - c]
    sec = float(np.mean(per))
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": round(1.0 / sec, 4),
        "unit": "sweeps/s",
        "cores": 1,
        "kind": "port",
        "sample": "stream 0, 2 registered sweeps of the benchmarked sensor (after 1 initialising sweep) vs the same frozen map; kd-tree build excluded",
        "seconds_per_sweep": round(sec, 4),
        "stage_seconds": {"features": round(stage[0] / 2, 4), "odometry": round(stage[1] / 2, 4), "registration": round(stage[2] / 2, 4)},
        "kdtree_build_seconds": round(t_build, 4),
        "host_cores_available": os.cpu_count(),
        "cpu_model": cpu_model,
        "compiler_flags": "g++ -O3 -march=native (oracle/liboracle_fast.so)",
    }


if __name__ == "__main__":
    main()
