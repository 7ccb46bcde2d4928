"""CPU: bench.py's own control flow, run end to end over stand-ins for torch's device side and for the library handle — the resident
window, its repeats, the PCIe-inclusive windows with the staging thread, the JSON line.  Nothing is measured here; the point is that the
driver's command cannot die of a Python-level mistake (an index past the staged sweeps, a misspelt key) that only a GPU box would otherwise
show.  What the stand-in pipeline checks on the way is the protocol bench.py promises: every step that runs was staged, steps run in order,
the steady-state window stages two more steps than it runs and drains the look-ahead before it closes, a slot of the streaming ring is
never re-staged while one of the eight steps in flight still owns it."""
import importlib
import json
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeTensor:
    def __init__(self, a):
        self.a = a

    def copy_(self, other, non_blocking=False):
        np.copyto(self.a, np.asarray(other.a if isinstance(other, FakeTensor) else other).reshape(self.a.shape))
        return self

    def pin_memory(self):
        return self

    def numpy(self):
        return self.a

    def cpu(self):
        return self

    def data_ptr(self):
        return self.a.ctypes.data

    def __getitem__(self, k):
        return FakeTensor(self.a[k])

    def __len__(self):
        return len(self.a)

    @property
    def shape(self):
        return self.a.shape


def fake_torch():
    t = types.ModuleType("torch")
    t.float32, t.uint8, t.float64 = np.float32, np.uint8, np.float64
    t.empty = lambda shape, dtype=np.float32, device=None: FakeTensor(np.zeros(shape, dtype))
    t.zeros = t.empty
    t.empty_like = lambda x: FakeTensor(np.zeros_like(x.a))
    t.from_numpy = lambda a: FakeTensor(a)
    t.frombuffer = lambda b, dtype=np.uint8: FakeTensor(np.frombuffer(b, dtype))
    t.device = lambda kind, idx=0: ("device", kind, idx)

    class Event:
        def __init__(self, enable_timing=False):
            self.cuda_event = 0

        def record(self):
            pass

        def synchronize(self):
            pass

        def elapsed_time(self, other):
            return 0.3

    def no_props(idx):
        raise RuntimeError("no device")
    t.cuda = types.SimpleNamespace(set_device=lambda i: None, synchronize=lambda: None, Event=Event, get_device_properties=no_props)

    class Scalar(FakeTensor):
        def item(self):
            return self.a.reshape(-1)[0]
    t.tensor = lambda v, dtype=np.float64, device=None: Scalar(np.array(v, dtype))
    # torch.distributed as the driver's launcher sets it up (one process here plays rank 0 of a larger world)
    d = types.ModuleType("torch.distributed")
    d.world = 1
    d.calls = []
    d.init_process_group = lambda backend, rank=0, world_size=1, device_id=None: (d.calls.append("init"), setattr(d, "world", world_size))[0]
    d.is_initialized = lambda: True
    d.get_world_size = lambda: d.world
    d.barrier = lambda: d.calls.append("barrier")
    d.broadcast = lambda tensor, src=0, async_op=False: types.SimpleNamespace(wait=lambda: None)
    d.all_reduce = lambda tensor, op=None: None
    d.ReduceOp = types.SimpleNamespace(MAX="max")
    d.destroy_process_group = lambda: d.calls.append("destroy")
    t.distributed = d
    return t


class FakeDist:
    """loamx.Dist: the library's own RCCL communicator"""
    ID_BYTES = 128
    made = []

    @staticmethod
    def unique_id():
        return bytes(range(128))

    def __init__(self, uid, rank, world, device=0):
        assert uid == bytes(range(128)) and 0 <= rank < world
        self.rank, self.world, self.broadcasts = rank, world, 0
        FakeDist.made.append(self)

    def broadcast_map(self, d_corner, n_corner, d_surf, n_surf, root=0, wait_event=0):
        assert d_surf == d_corner + 16 * n_corner and root == 0
        self.broadcasts += 1
        return 0

    def barrier(self):
        pass

    def allgather_results(self, poses6, iters_flags=None, batch=None):
        p = np.asarray(poses6, np.float32).reshape(-1, 6)
        assert batch == self.world * len(p) and np.asarray(iters_flags).shape == (len(p), 2)
        return np.tile(p, (self.world, 1)), np.tile(np.asarray(iters_flags, np.int32), (self.world, 1)), np.full(self.world, len(p), np.uint32)

    def comm_count(self):
        return self.world

    def allgather_counts(self, n):
        return np.full(self.world, n, np.uint32)

    def gatherv(self, words, root=0):
        # every other rank is played by a copy of this rank's message (the layout is the library's own: loamx.dist_pack_clouds is real)
        w = np.asarray(words, np.uint32)
        self.gathers = getattr(self, "gathers", 0) + 1
        return ([w if len(w) else None for _ in range(self.world)] if self.rank == root else None), np.full(self.world, len(w), np.uint32)


class FakePipeline:
    """the calls bench.py makes on loamx.Pipeline, with the ordering rules of include/loamx.h asserted"""
    instances = []

    def __init__(self, n_streams, **cfg):
        self.ns, self.h = n_streams, id(self)
        self.uploaded = 0           # steps staged up front (loamx_pipeline_upload)
        self.staged = 0             # steps staged one at a time (loamx_pipeline_stage_step)
        self.last = -1
        self.drained_at = None
        self.closed = False
        self.async_dl = False
        self.downloads = 0
        FakePipeline.instances.append(self)

    def set_frozen_device(self, *a):
        pass

    def set_state(self, k, aft=None):
        assert 0 <= k < self.ns and aft is not None and len(aft) == 6

    def upload(self, sweeps):
        assert all(len(row) == self.ns for row in sweeps)
        self.uploaded = len(sweeps)

    def set_timing(self, on, per_launch=True):
        pass

    def enable_async_downloads(self):
        self.async_dl = True

    def step(self, t):
        assert not self.closed and t == self.last + 1, "steps run in order"
        assert t < max(self.uploaded, self.staged), "step beyond the staged sweeps"
        self.last = t
        return 1 if t == 0 else 0   # the first sweep of a stream only initialises the odometry

    def swap_frozen(self):
        sw = getattr(self, "staged_maps", 0) > getattr(self, "swapped", 0)
        self.swapped = getattr(self, "swapped", 0) + (1 if sw else 0)
        return sw

    def stage_frozen_device(self, d_corner, n_corner, d_surf, n_surf, ev=0):
        assert d_surf == d_corner + 16 * n_corner and n_corner > 0 and n_surf > 0
        self.staged_maps = getattr(self, "staged_maps", 0) + 1
        self.map_sizes = getattr(self, "map_sizes", []) + [(n_corner, n_surf)]

    def last_clouds(self, stream, capacity):
        assert self.last >= 1 and 0 <= stream < self.ns
        rng = np.random.default_rng(1000 * self.last + stream)
        return rng.normal(size=(40, 4)).astype(np.float32), rng.normal(size=(90, 4)).astype(np.float32)

    def drain_lookahead(self):
        self.drains = getattr(self, "drains", [])
        self.drains.append(self.last)
        self.drained_at = self.last
        return min(self.last + self.lookahead_depth(), max(self.uploaded, self.staged) - 1)

    def lookahead_depth(self):
        return 6

    def timing(self):
        return dict(features_ms=0.2, odometry_ms=0.5, registration_ms=0.4, step_ms=0.4, residual_ms=0.18, residual_launches=3,
                    query_iterations=3 * 1000 * self.ns, queries=1000 * self.ns, run_ms=0.4)

    def odom_launch_timing(self):
        return dict(lm_ms=0.4, lm_noop_ms=0.02, corr_ms=0.3, corr_noop_ms=0.02, lm_launches=10, lm_noop_launches=4, lm_iterations=44, corr_launches=10,
                    corr_noop_launches=4, lm_bytes=10 * 48 * 2304 * self.ns, corr_features=10 * 2304 * self.ns)

    def get(self, k):
        z = np.zeros(6, np.float32)
        return z, z, z, dict(map_iterations=3, mapped=1, odom_iterations=7, odom_sel=100, map_sel=500, corner_ds=100, surf_ds=900, degenerate=0)

    def wait_downloads(self):
        pass

    def download_counts(self):
        return self.downloads, 0

    def close(self):
        self.closed = True


class FakeLib:
    """the two C entry points bench.py calls directly in the PCIe window (+ by name, entry points of the real library: host-side layout
    functions that need no GPU)"""
    def __init__(self, loamx, real=()):
        self.loamx = loamx
        if real:
            import ctypes
            L = ctypes.CDLL(loamx.LIB_PATH)
            L.loamx_last_error.restype = ctypes.c_char_p
            L.loamx_dist_pack_clouds.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
            L.loamx_dist_unpack_clouds_header.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
            L.loamx_dist_unpack_clouds_stream.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            for n in real:
                setattr(self, n, getattr(L, n))

    def _pipe(self, h):
        return next(p for p in FakePipeline.instances if p.h == h)

    def loamx_pipeline_stage_step(self, h, t, clouds, rings, nrings):
        p = self._pipe(h)
        assert t == p.staged, "steps are staged in order"
        assert t < 8 or p.last >= t - 8, "stage_step(t) needs step(t - 8) to have returned"
        assert sum(int(x) for x in np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint32 * int(nrings[0])).from_address(rings[0]))) == clouds[0].count
        p.staged = t + 1
        return 0

    def loamx_pipeline_download_step_async(self, h, clouds, n):
        p = self._pipe(h)
        assert p.async_dl and n == p.ns
        p.downloads += 1
        return 0

    def loamx_last_error(self):
        return b""

    def loamx_build_info(self):
        return b"abi=6;diag=0;rccl=1;roctx=1"


@pytest.mark.parametrize("argv", [["--steps", "3", "--warmup", "1", "--streams", "2", "--sensor", "VLP-16", "--map-points", "2000", "--no-cpu-baseline",
                                   "--repeat", "2", "--no-side-configs"],
                                  ["--steps", "3", "--warmup", "1", "--streams", "2", "--sensor", "VLP-16", "--map-points", "2000", "--no-cpu-baseline",
                                   "--repeat", "2", "--handles", "2", "--no-side-configs"]])
def test_bench_main_runs_over_stand_ins(monkeypatch, capsys, argv):
    H = int(argv[argv.index("--handles") + 1]) if "--handles" in argv else 1
    from loam_velodyne_amd import loamx
    monkeypatch.setitem(sys.modules, "torch", fake_torch())
    monkeypatch.setattr(loamx, "Pipeline", FakePipeline)
    monkeypatch.setattr(loamx, "lib", lambda: FakeLib(loamx))
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    FakePipeline.instances.clear()
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    bench.main()
    line = [l for l in capsys.readouterr().out.strip().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    K, W, ns = 3, 1, 2
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "value_median", "value_min", "value_max"):
        assert key in out, key
    assert out["steps"] == K and out["warmup"] == W and out["n_gpus"] == 1 and out["config"]["streams_per_gpu"] == ns
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "dominant_by", "peak_copy_ceiling"}
    if out["roofline"]["kernel"].startswith("loamx::k_odom_lm"):   # (the committed kernel stats name the dominant kernel)
        assert set(out["roofline"]["latency_model"]) >= {"us_per_iteration", "floor_us", "source"} and "k_gn_iter" in out["roofline"]
    assert "env_overrides" in out["config"] and out["config"]["library_build"]["diag"] == "0"
    assert out["value_repeats"] == 2 and out["config"]["handles_per_gpu"] == H
    if H == 1:
        assert len(out["pcie_inclusive"]["value_windows"]) == 3
    else:   # (the PCIe-inclusive window is a single-handle measurement)
        assert out["pcie_inclusive"] is None
    # the windows: every resident one staged K + 1 + W + 6 steps (the staged batches' look-ahead depth), ran 1 + W + K of them, and drained
    # the look-ahead after its last step; the PCIe ones staged the same number one at a time
    resident = [p for p in FakePipeline.instances if p.uploaded]
    streaming = [p for p in FakePipeline.instances if p.staged]
    assert len(resident) == 2 * H and len(streaming) == (3 if H == 1 else 0)
    assert all(p.ns == ns // H for p in resident)   # the streams are dealt over the handles, each driven by its own host thread
    for p in resident:   # (drained when the window opens — after the last warm-up step — and before it closes)
        assert p.uploaded == 1 + W + K + 6 and p.last == W + K and p.drains == [W, W + K] and p.closed
    for p in streaming:
        assert p.staged == 1 + W + K + 6 and p.last == W + K and p.drains == [W, W + K] and p.downloads == W + K and p.closed


def test_bench_main_as_rank_0_of_two(monkeypatch, capsys):
    """the multi-GPU path of bench.py (process group, the library's own communicator for the map broadcast and the result gather,
    max-over-ranks timing) — never executed on hardware so far: at least its Python must hold"""
    from loam_velodyne_amd import loamx
    ft = fake_torch()
    monkeypatch.setitem(sys.modules, "torch", ft)
    monkeypatch.setitem(sys.modules, "torch.distributed", ft.distributed)
    monkeypatch.setattr(loamx, "Pipeline", FakePipeline)
    monkeypatch.setattr(loamx, "Dist", FakeDist)
    monkeypatch.setattr(loamx, "lib", lambda: FakeLib(loamx))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--streams", "2", "--sensor", "VLP-16", "--map-points", "2000",
                                      "--repeat", "2"])
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    FakePipeline.instances.clear()
    FakeDist.made.clear()
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    bench.main()
    out = json.loads([l for l in capsys.readouterr().out.strip().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "cpu_baseline" not in out
    assert out["config"]["rccl_ranks"] == 2 and out["config"]["results_gathered"] == 4 and "native" in out["config"]["map_broadcast_via"]
    assert len(FakeDist.made) == 1 and FakeDist.made[0].broadcasts == 2          # (communicator warm-up + the timed broadcast)
    assert ft.distributed.calls[0] == "init" and ft.distributed.calls[-1] == "destroy"
    # value counts both ranks' sweeps over this rank's clock (the stand-in's max-over-ranks is the identity)
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 2 * 2) < 0.05 * 4


def test_pose_error_and_roofline_kernels_helpers():
    """bench.py's parity figure (BASELINE.json metric part 3) and the extra roofline rows, on synthetic inputs"""
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    rng = np.random.default_rng(0)
    orc, gpu = [], []
    for t in range(1, 8):
        ts, aft = rng.normal(size=6).astype(np.float32), rng.normal(size=6).astype(np.float32)
        orc.append((t, ts, aft, 6, 3))
        for s in range(2):   # stream 1 is ignored
            d = np.float32(2e-6 if s == 0 else 1.0)
            gpu.append((t, s, ts + d, aft - d, 6, 3 if t != 4 else 4))
    pe = bench.pose_error(gpu, orc, stream=0)
    assert pe["sweeps"] == 7 and pe["within_bar"] and 1e-6 < pe["mapped_pose"]["max_m"] < 1e-5 and pe["odometry_iterations_equal"] == 7
    assert pe["mapping_iterations_equal"] == 6   # (sweep 4 was made to differ)
    gpu_bad = [(t, s, ts + np.float32(1e-3), aft, oi, mi) for (t, s, ts, aft, oi, mi) in gpu]
    assert not bench.pose_error(gpu_bad, orc, stream=0)["within_bar"]
    assert bench.pose_error([], orc) is None
    rows = bench.roofline_kernels({"kernel": "loamx::k_gn_iter"}, 8)
    assert {r["kernel"] for r in rows} >= {"loamx::k_odom_corr_grid", "loamx::k_odom_lm<1>", "loamx::k_vb_reduce", "loamx::k_feat_ring"}
    for r in rows:
        assert r["algorithmic_bytes_per_launch"] > 0 and set(r) >= {"avg_launch_us", "frac", "traffic", "model", "source"}


class FakeMapping:
    """loamx.LaserMapping as the epoch's accumulator: counts what is inserted"""
    made = []

    def __init__(self, **cfg):
        self.c = self.s = None
        self.inserted = []
        FakeMapping.made.append(self)

    def load_cubes(self, cm, sm):
        self.c, self.s = np.asarray(cm, np.float32).copy(), np.asarray(sm, np.float32).copy()

    def insert(self, corner, surf, pose6):
        assert corner.shape[1] == 4 and surf.shape[1] == 4 and np.asarray(pose6).shape == (6,)
        self.inserted.append((len(corner), len(surf)))
        self.c = np.concatenate([self.c, corner[:3]])   # the merged map grows
        self.s = np.concatenate([self.s, surf[:5]])
        return 0

    def cubes(self, which):
        return self.c if which == "corner" else self.s


def test_bench_epoch_merge_as_rank_0_of_eight(monkeypatch, capsys):
    """bench.py --map-epoch-steps E --epoch-merge as rank 0 of EIGHT: at every epoch boundary the ranks' sweeps are packed with the library's
    own layout, gathered to rank 0 (the stand-in communicator plays the other seven ranks), inserted into the accumulator, and the merged
    map — of a new size every epoch — is what is broadcast and staged.  The multi-GPU epoch has never run on hardware: its control flow
    at least runs here, with the real pack / unpack code under it."""
    from loam_velodyne_amd import loamx
    ft = fake_torch()
    monkeypatch.setitem(sys.modules, "torch", ft)
    monkeypatch.setitem(sys.modules, "torch.distributed", ft.distributed)
    monkeypatch.setattr(loamx, "Pipeline", FakePipeline)
    monkeypatch.setattr(loamx, "Dist", FakeDist)
    monkeypatch.setattr(loamx, "LaserMapping", FakeMapping)
    monkeypatch.setattr(loamx, "lib", lambda: FakeLib(loamx, real=("loamx_dist_pack_clouds", "loamx_dist_unpack_clouds_header", "loamx_dist_unpack_clouds_stream")))
    K, W, ns, E = 6, 1, 2, 2
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", str(K), "--warmup", str(W), "--streams", str(ns), "--sensor", "VLP-16", "--map-points", "2000",
                                      "--repeat", "1", "--no-pcie", "--map-epoch-steps", str(E), "--epoch-merge"])
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "8")
    FakePipeline.instances.clear()
    FakeDist.made.clear()
    FakeMapping.made.clear()
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    bench.main()
    out = json.loads([l for l in capsys.readouterr().out.strip().splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["config"]["rccl_ranks"] == 8 and out["config"]["map_epoch_steps"] == E
    boundaries = K // E - 1                       # the first epoch has nothing to merge yet
    merged = out["config"]["map_epoch_merge"]["merged_sweeps_on_rank0"]
    assert merged == boundaries * 8 * ns, (merged, boundaries)
    acc = FakeMapping.made[-1]
    assert len(acc.inserted) == merged and FakeDist.made[0].gathers == boundaries
    sizes = FakePipeline.instances[-1].map_sizes
    assert len(sizes) >= K // E and sizes[-1][0] > sizes[0][0] and sizes[-1][1] > sizes[0][1]   # the staged map grew with the merged sweeps


def test_bench_async_epoch_merge_as_rank_0_of_two(monkeypatch, capsys):
    """bench.py --map-epoch-steps E --epoch-merge-async as rank 0 of two: the accumulator on a worker thread with a communicator of its own;
    the stepping thread only hands sweeps over when the worker is idle and stages a merged map once one has arrived.  With the stand-ins the
    worker is quick, so merged maps do arrive inside the window: they were staged (sizes growing) and swapped in at epoch boundaries."""
    from loam_velodyne_amd import loamx
    ft = fake_torch()
    ft.cuda.Event.cuda_event = 0
    monkeypatch.setitem(sys.modules, "torch", ft)
    monkeypatch.setitem(sys.modules, "torch.distributed", ft.distributed)
    monkeypatch.setattr(loamx, "Pipeline", FakePipeline)
    monkeypatch.setattr(loamx, "Dist", FakeDist)
    monkeypatch.setattr(loamx, "LaserMapping", FakeMapping)
    monkeypatch.setattr(loamx, "lib", lambda: FakeLib(loamx, real=("loamx_dist_pack_clouds", "loamx_dist_unpack_clouds_header", "loamx_dist_unpack_clouds_stream")))
    K, W, ns, E = 12, 1, 2, 2
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", str(K), "--warmup", str(W), "--streams", str(ns), "--sensor", "VLP-16", "--map-points", "2000",
                                      "--repeat", "1", "--no-pcie", "--map-epoch-steps", str(E), "--epoch-merge-async"])
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    FakePipeline.instances.clear()
    FakeDist.made.clear()
    FakeMapping.made.clear()
    orig_step = FakePipeline.step

    def slow_step(self, t):   # (give the worker thread a chance between two steps, as a real step's 0.4 ms does)
        import time
        time.sleep(0.01)
        return orig_step(self, t)
    monkeypatch.setattr(FakePipeline, "step", slow_step)
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    bench.main()
    out = json.loads([l for l in capsys.readouterr().out.strip().splitlines() if l.startswith("{")][-1])
    em = out["config"]["map_epoch_merge"]
    assert em["mode"].startswith("asynchronous") and em["merge_jobs_completed_in_window"] >= 1 and em["merged_sweeps_on_rank0"] >= 2 * ns
    assert len(FakeDist.made) == 2                      # the stepping thread's communicator and the worker's own
    assert FakeDist.made[1].gathers == em["merge_jobs_completed_in_window"] or FakeDist.made[1].gathers == em["merge_jobs_completed_in_window"] + 1
    assert getattr(FakeDist.made[0], "gathers", 0) == 0  # nothing of the merge on the stepping thread's communicator
    sizes = FakePipeline.instances[-1].map_sizes
    assert len(sizes) >= 2 and sizes[-1][0] > sizes[0][0]   # (the untimed first stage, then merged maps of growing size)
    assert em["merged_maps_swapped_in"] >= 1


def _canned_live(sensor, M, K, W, cpu=True, nodes=True, within=True):
    """what bench.live_block returns, as far as the line's assembly reads it"""
    return {"metric": f"sweeps/sec (sequential SLAM: {sensor})", "value": 1500.0, "unit": "sweeps/s", "ms_per_step": 0.66, "steps": K, "warmup": W,
            "config": {"workload": sensor}, "roofline": {"kernel": "loamx::k_gn_iter", "frac": 0.003},
            "cpu_baseline": {"value": 25.0, "unit": "sweeps/s", "cores": 1, "kind": "port"},
            "pose_err_vs_oracle": {"mapped_pose": {"max_m": 4e-4, "max_rad": 1e-6, "rmse_m": 1e-4}, "bar_free_running": 1.4e-3, "within_bar": within}}


@pytest.mark.parametrize("within", [True, False])
def test_default_line_carries_every_single_gpu_configuration(monkeypatch, capsys, within):
    """the driver runs `bench.py --gpus 1` only: the line must carry BASELINE configs[1], [2] and configs[4]'s one-GPU point as blocks, their
    headline figures flat in `config` and once more in the line's LAST key — and the run must exit with status 3 when any block's parity is
    outside its bar (after the line has been printed)"""
    from loam_velodyne_amd import loamx
    monkeypatch.setitem(sys.modules, "torch", fake_torch())
    monkeypatch.setattr(loamx, "Pipeline", FakePipeline)
    monkeypatch.setattr(loamx, "lib", lambda: FakeLib(loamx))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "3", "--warmup", "1", "--streams", "2", "--sensor", "VLP-16", "--map-points", "2000", "--map2-points", "4000",
                                      "--no-cpu-baseline", "--repeat", "1", "--no-pcie"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    FakePipeline.instances.clear()
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(bench, "side_live", lambda *a, **kw: _canned_live(*a, **kw, within=within))
    if within:
        bench.main()
    else:
        with pytest.raises(SystemExit) as e:
            bench.main()
        assert e.value.code == 3
    line = [l for l in capsys.readouterr().out.strip().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert list(out)[-1] == "summary"
    for key in ("map_2m", "live_vlp16", "live_hdl32"):
        assert key in out and "error" not in out[key], out.get(key)
    assert out["map_2m"]["workload"].startswith("BASELINE configs[4]") and "4000-pt" in out["map_2m"]["workload"]
    for k in ("live_vlp16_sweeps_per_s", "live_vlp16_ms_per_sweep", "live_vlp16_pose_max_m", "live_vlp16_within_bar", "live_hdl32_sweeps_per_s",
              "map_2m_sweeps_per_s", "map_2m_ms_per_step", "all_within_bar"):
        assert k in out["summary"] and out["config"][k] == out["summary"][k], k
    assert out["summary"]["all_within_bar"] is within
    # the 2 M block ran its windows against its own map and put the first map back
    assert out["config"]["map_points"] == 2000


def _rows(n, d_at=None, counts_at=None):
    """(gpu rows, oracle rows): identical chains except a mapped-pose difference d at sweep index d_at (and other counts there)"""
    g, o = [], []
    for t in range(1, n + 1):
        ts = np.full(6, 0.01 * t, np.float32)
        aft = np.full(6, 0.02 * t, np.float32)
        cnt = (1800, 11000, 1300, 12000)
        ag, cg = aft.copy(), cnt
        if d_at is not None and t == d_at[0]:
            ag[3] += np.float32(d_at[1])
            if counts_at:
                cg = (1800, 10999, 1299, 12000)
        g.append((t, 0, ts, ag, 6, 3, cg))
        o.append((t, ts, aft, 6, 3, cnt))
    return g, o


def test_parity_gate_rules():
    """bench.pose_error's bar: flat 1e-4 without an envelope; with one, max(1e-4, envelope) — and a sweep beyond that only as an explained
    threshold flip (per step from identical state <= 1e-6 there and a discrete count differing), the distribution not above the envelope's"""
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    g, o = _rows(50)
    assert bench.pose_error(g, o)["within_bar"] is True
    g, o = _rows(50, d_at=(20, 3e-4))
    pe = bench.pose_error(g, o)
    assert pe["within_bar"] is False and pe["mapped_pose"]["max_at"]["sweep"] == 20 and pe["mapped_pose"]["max_at"]["component"] == "x"
    env = {"max_m": 1.7e-4, "max_rad": 1e-5, "rmse_m": 5e-5, "p99_m": 2e-4, "sweeps_above_1e-4": 3, "odometry_sum_max_m": 5e-4, "pairs": {}}
    ps_ok = {"max_m": 7e-5, "max_rad": 1e-6, "by_sweep": {t: 1e-8 for t in range(1, 51)}}
    # beyond the envelope, per step fine, but no count differs: not explained
    assert bench.pose_error(g, o, envelope=env, per_step=ps_ok)["within_bar"] is False
    # ... with a differing count at that sweep: an explained flip
    g, o = _rows(50, d_at=(20, 3e-4), counts_at=True)
    pe = bench.pose_error(g, o, envelope=env, per_step=ps_ok)
    assert pe["within_bar"] is True and pe["sweeps_outside_bar_free_running"][0]["explained_as_threshold_flip"] is True and pe["bar_free_running"] == pytest.approx(1.7e-4)
    assert pe["counts_equal"]["map_sel"] == 49 and pe["counts_equal"]["odom_sel"] == 50
    # the same flip, but the device differs from the oracle on identical inputs at that sweep: not a flip of the chain, a difference of the arithmetic
    ps_bad = dict(ps_ok, by_sweep={**ps_ok["by_sweep"], 20: 5e-5})
    assert bench.pose_error(g, o, envelope=env, per_step=ps_bad)["within_bar"] is False
    # per step from identical state above 1e-4 anywhere fails whatever the envelope says
    assert bench.pose_error(*_rows(50), envelope=env, per_step=dict(ps_ok, max_m=2e-4))["within_bar"] is False
    # inside the envelope: fine without any explanation
    g, o = _rows(50, d_at=(20, 1.5e-4))
    assert bench.pose_error(g, o, envelope=env, per_step=ps_ok)["within_bar"] is True


def test_reference_envelope_block():
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    base = np.zeros((30, 13)); base[:, 0] = np.arange(1, 31)
    fast = base.copy(); fast[7, 10] += 1.2e-4
    ref = base.copy()
    alt = base.copy(); alt[9, 12] -= 2.0e-4; alt[3, 8] += 3e-6
    env = bench.reference_envelope({"oracle_fast": fast, "ref": ref, "ref_map_alt": alt}, base)
    assert env["pairs"]["ref"]["mapped_max_m"] == 0.0                      # the pin: reported, not part of the envelope
    assert env["max_m"] == pytest.approx(2.0e-4) and env["pairs"]["ref_map_alt"]["mapped_max_at_sweep"] == 10
    assert env["sweeps_above_1e-4"] == 1 and env["max_rad"] == pytest.approx(3e-6)
