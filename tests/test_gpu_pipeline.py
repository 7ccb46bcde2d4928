"""GPU: the streaming pipeline (features -> odometry -> frozen-map registration for n streams) vs the oracle chain, the
committed golden fixture, and the C++ adapter classes that mirror the reference interface."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_py as op
from conftest import GOLDEN, POSE_TOL, ROOT
from loam_velodyne_amd import loamx, synth

pytestmark = pytest.mark.gpu


def test_golden_pipeline():
    g = np.load(os.path.join(GOLDEN, "pipeline_vlp16.npz"))
    pipe = loamx.Pipeline(2)
    pipe.set_frozen(g["corner_map"], g["surf_map"])
    for s in range(2):
        pipe.set_state(s, aft=g[f"start_{s}"])
    pipe.upload([[(g[f"points_{s}_{t}"], g[f"rings_{s}_{t}"]) for s in range(2)] for t in range(4)])
    for t in range(4):
        rc = pipe.step(t)
        assert rc == (loamx.SKIPPED if t == 0 else loamx.OK)
        for s in range(2):
            _, ts, aft, st = pipe.get(s)
            assert np.abs(ts - g[f"sum_{s}"][t]).max() < POSE_TOL
            assert np.abs(aft - g[f"aft_{s}"][t]).max() < POSE_TOL
            assert st["mapped"] == (1 if t > 0 else 0)


def test_streams_are_independent(orc, small_world):
    """A stream's results do not depend on which other streams share the batch (bit-identical)."""
    cm, sm = small_world.make_map(40000)
    T = 3

    def stream(s):
        poses = synth.trajectory(T, start=(1.0 * s, 0.0, 2.0 * s))
        return [synth.make_sweep(small_world, "VLP-16", poses[t], poses[t + 1], seed=10 * s + t, az_steps=800) for t in range(T)]
    data = [stream(s) for s in range(3)]

    def run(ids):
        p = loamx.Pipeline(len(ids))
        p.set_frozen(cm, sm)
        for k, s in enumerate(ids):
            p.set_state(k, aft=np.array([0, 0, 0, 1.0 * s, 0, 2.0 * s], np.float32))
        p.upload([[(data[s][t].points, data[s][t].ring_sizes) for s in ids] for t in range(T)])
        for t in range(T):
            p.step(t)
        return [p.get(k)[:3] for k in range(len(ids))]
    all3 = run([0, 1, 2])
    alone = run([1])
    swapped = run([2, 1])
    for a, b in ((all3[1], alone[0]), (all3[1], swapped[1]), (all3[2], swapped[0])):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_lookahead_is_transparent(small_world):
    """The software pipeline across steps (registration t || odometry t+1 || features t+2 on separate HIP streams and a
    host thread) must not change any result: bit-identical to the strictly sequential execution."""
    cm, sm = small_world.make_map(40000)
    T, ns = 9, 3
    data = []
    for s in range(ns):
        poses = synth.trajectory(T, start=(1.0 * s, 0.0, 2.0 * s))
        data.append([synth.make_sweep(small_world, "VLP-16", poses[t], poses[t + 1], seed=30 * s + t, az_steps=800) for t in range(T)])

    def run(lookahead):
        p = loamx.Pipeline(ns)
        p.set_lookahead(lookahead)
        p.set_frozen(cm, sm)
        for s in range(ns):
            p.set_state(s, aft=np.array([0, 0, 0, 1.0 * s, 0, 2.0 * s], np.float32))
        p.upload([[(data[s][t].points, data[s][t].ring_sizes) for s in range(ns)] for t in range(T)])
        out = []
        for t in range(T):
            rc = p.step(t)
            if lookahead:   # the look-ahead runs up to lookahead_depth() steps beyond the step that has returned — and no further
                assert p.lookahead_depth() == 6
                assert p.drain_lookahead() == min(t + 6, T - 1)
            else:
                assert p.lookahead_depth() == 0 and p.drain_lookahead() == t
            out.append((rc, [p.get(s) for s in range(ns)]))
        return out
    a, b = run(True), run(False)
    for (rca, ga), (rcb, gb) in zip(a, b):
        assert rca == rcb
        for (tra, tsa, afa, sta), (trb, tsb, afb, stb) in zip(ga, gb):
            assert np.array_equal(tra, trb) and np.array_equal(tsa, tsb) and np.array_equal(afa, afb) and sta == stb


def test_launch_pairs_as_needed_change_nothing(small_world, monkeypatch):
    """Round 6: the odometry's launch pairs (correspondences + five iterations) are enqueued as they turn out to be needed — one pair behind
    the device, decided from the pinned mirror — instead of all five up front (the reference leaves its loop when the stop test fires,
    BasicLaserOdometry.cpp:613-620).  A pair that is not enqueued would have returned at its first instruction, so LOAMX_ODOM_PAIRS = all
    (the fixed five), lag (default: one launch ahead of need), lag2 (one pair ahead) and exact (none) must give bit-identical transforms, iteration counts
    and mapped poses — in the batched pipeline and through the single-stream odometry handle."""
    cm, sm = small_world.make_map(40000)
    T, ns = 7, 3
    data = []
    for s in range(ns):
        poses = synth.trajectory(T, start=(1.0 * s, 0.0, 2.0 * s))
        data.append([synth.make_sweep(small_world, "VLP-16", poses[t], poses[t + 1], seed=40 * s + t, az_steps=800) for t in range(T)])

    def run(mode):
        if mode is None:
            monkeypatch.delenv("LOAMX_ODOM_PAIRS", raising=False)
        else:
            monkeypatch.setenv("LOAMX_ODOM_PAIRS", mode)
        p = loamx.Pipeline(ns)
        p.set_frozen(cm, sm)
        for s in range(ns):
            p.set_state(s, aft=np.array([0, 0, 0, 1.0 * s, 0, 2.0 * s], np.float32))
        p.upload([[(data[s][t].points, data[s][t].ring_sizes) for s in range(ns)] for t in range(T)])
        out = []
        for t in range(T):
            rc = p.step(t)
            out.append((rc, [p.get(s) for s in range(ns)]))
        p.close()
        # the single-stream handle (the sequential-SLAM entry points) on stream 0's sweeps
        sr, od = loamx.ScanRegistration(), loamx.LaserOdometry()
        single = []
        for t in range(T):
            od.process(sr.process(data[0][t].points, data[0][t].ring_sizes))
            single.append((np.array(od.transform), np.array(od.transform_sum), od.stats()))
        sr.close(); od.close()
        return out, single
    ref_out, ref_single = run("all")
    assert max(st["odom_iterations"] for _, g in ref_out for (_, _, _, st) in g) > 5   # (more than one pair was needed somewhere)
    for mode in (None, "lag", "lag2", "exact"):
        out, single = run(mode)
        for (rca, ga), (rcb, gb) in zip(ref_out, out):
            assert rca == rcb
            for (tra, tsa, afa, sta), (trb, tsb, afb, stb) in zip(ga, gb):
                assert np.array_equal(tra, trb) and np.array_equal(tsa, tsb) and np.array_equal(afa, afb) and sta == stb, mode
        for (ta, sa, xa), (tb, sb, xb) in zip(ref_single, single):
            assert np.array_equal(ta, tb) and np.array_equal(sa, sb) and xa == xb, mode


def test_lookahead_toggled_and_state_set_mid_run(small_world):
    """ADVICE.md (round 3): loamx_pipeline_set_lookahead(0) and loamx_pipeline_set_state in the MIDDLE of a run, right after step() has
    returned and while the look-ahead is still working on the next steps, must leave the odometry chain where it really is — the old
    code re-ran a step the worker had just finished (transformSum integrated twice).  A run with the look-ahead switched off after
    step 2 and on again after step 4 equals the undisturbed run bit for bit; a set_state(aft=...) with the value the stream already
    has changes nothing either."""
    cm, sm = small_world.make_map(40000)
    T, ns = 8, 3
    data = []
    for s in range(ns):
        poses = synth.trajectory(T, start=(1.0 * s, 0.0, 2.0 * s))
        data.append([synth.make_sweep(small_world, "VLP-16", poses[t], poses[t + 1], seed=50 * s + t, az_steps=800) for t in range(T)])

    def run(disturb):
        p = loamx.Pipeline(ns)
        p.set_frozen(cm, sm)
        for s in range(ns):
            p.set_state(s, aft=np.array([0, 0, 0, 1.0 * s, 0, 2.0 * s], np.float32))
        p.upload([[(data[s][t].points, data[s][t].ring_sizes) for s in range(ns)] for t in range(T)])
        out = []
        for t in range(T):
            p.step(t)
            out.append([p.get(s) for s in range(ns)])
            if disturb and t == 2:
                p.set_lookahead(False)      # (the look-ahead is inside O(3) / O(4) right now)
            if disturb and t == 4:
                p.set_lookahead(True)
            if disturb and t == 5:
                for s in range(ns):
                    p.set_state(s, aft=out[-1][s][2])   # the value it has: a no-op that parks the chain mid-flight
        p.close()
        return out
    a, b = run(False), run(True)
    for t in range(T):
        for s in range(ns):
            for x, y in zip(a[t][s][:3], b[t][s][:3]):
                assert np.array_equal(x, y), (t, s)
            assert a[t][s][3] == b[t][s][3], (t, s)


def test_cpp_adapter_matches(orc, small_world, tmp_path):
    """The header-only C++ classes with the reference's member names (loam_velodyne_amd/adapter) drive the same library:
    scan registration -> odometry -> mapping exactly as the reference's node loop wires them."""
    exe = os.path.join(ROOT, "loam_velodyne_amd", "adapter", "adapter_test")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe), "-s"], check=True)
    n = 5
    poses = synth.trajectory(n)
    sws = [synth.make_sweep(small_world, "VLP-16", poses[k], poses[k + 1], seed=k, az_steps=900) for k in range(n)]
    path = tmp_path / "sweeps.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("i", n))
        for sw in sws:
            f.write(struct.pack("i", len(sw.ring_sizes)))
            f.write(np.asarray(sw.ring_sizes, np.int32).tobytes())
            f.write(np.ascontiguousarray(sw.points, np.float32).tobytes())
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = np.array([[float(v) for v in line.split()[1:]] for line in out.stdout.strip().splitlines()])
    assert rows.shape == (n, 12)
    # the same chain through the Python binding of the same C-ABI
    gsr, god, gmp = loamx.ScanRegistration(), loamx.LaserOdometry(), loamx.LaserMapping()
    osr, ood, omp = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc)
    for k, sw in enumerate(sws):
        f = gsr.process(sw.points, sw.ring_sizes)
        god.process(f)
        lc, ls = god.last_clouds()
        gmp.update_odometry(god.transform_sum)
        gmp.process(lc, ls, god.transform_to_end(f["full"]))
        assert np.allclose(rows[k, :6], god.transform_sum, atol=1e-6)
        assert np.allclose(rows[k, 6:], gmp.transform("aft"), atol=1e-6)
        ood.set_features(osr.process(sw.points, sw.ring_sizes))
        ood.process()
        omp.set_inputs(ood.last_corner(), ood.last_surf(), ood.full_to_end(), ood.transform_sum)
        omp.process()
        assert np.abs(rows[k, :6] - ood.transform_sum).max() < POSE_TOL
    assert np.abs(rows[-1, 6:] - omp.transform("aft")).max() < 2e-3     # free-running live map: see test_gpu_mapping


def test_full_size_hdl64_pipeline_vs_oracle(orc):
    """BASELINE configs[3] shapes — HDL-64E sweeps (64 x 2048 = 131,072 points), 1,000,000-point frozen map — through the whole
    streaming path (features -> odometry -> registration, look-ahead on) for 2 streams x 3 sweeps against the oracle chain."""
    world = synth.World(half_extent=125.0)
    cm, sm = world.make_map(1000000)
    NS, T = 2, 3
    starts = [(3.0 * s - 1.0, 0.0, 4.0 * s) for s in range(NS)]
    data = []
    for s in range(NS):
        poses = synth.trajectory(T, start=starts[s])
        data.append([synth.make_sweep(world, "HDL-64E", poses[t], poses[t + 1], seed=77 * s + t) for t in range(T)])
    assert data[0][0].points.shape == (131072, 4)
    pipe = loamx.Pipeline(NS)
    pipe.set_frozen(cm, sm)
    for s in range(NS):
        pipe.set_state(s, aft=np.array([0, 0, 0, *starts[s]], np.float32))
    pipe.upload([[(data[s][t].points, data[s][t].ring_sizes) for s in range(NS)] for t in range(T)])
    got = []
    for t in range(T):
        assert pipe.step(t) == (loamx.SKIPPED if t == 0 else loamx.OK)
        got.append([pipe.get(s) for s in range(NS)])
    for s in range(NS):
        osr, ood, omp = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc)
        omp.set_frozen(cm, sm)
        omp.set_transform("aft", np.array([0, 0, 0, *starts[s]], np.float32))
        for t in range(T):
            ood.set_features(osr.process(data[s][t].points, data[s][t].ring_sizes))
            ood.process()
            if t > 0:
                omp.set_transform("sum", ood.transform_sum)
                omp.register_frozen(ood.last_corner(), ood.last_surf(), omp.associate())
            tr, ts, aft, st = got[t][s]
            assert np.abs(ts - ood.transform_sum).max() < POSE_TOL, (s, t)
            assert np.abs(aft - omp.transform("aft")).max() < POSE_TOL, (s, t)
            if t > 0:
                assert st["odom_iterations"] == ood.stats()["iterations"] and st["map_iterations"] == omp.stats()["iterations"]
        # sanity against ground truth (not the parity check): after two moving sweeps the registered position is near the
        # sensor position at the end of the last sweep (LOAM with a 25 / 10 iteration budget, not a converged optimum)
        gt = synth.trajectory(T, start=starts[s])[T]
        assert np.abs(got[T - 1][s][2][3:] - gt[3:]).max() < 0.3


def _bench_scale_chain(args):
    """worker process: the oracle chain of one stream over its sweeps (CPU) -> per sweep (transformSum, mapped pose, iteration counts)"""
    s, start, cm, sm, sweeps = args
    import oracle_py as op_
    o = op_.Oracle()
    osr, ood, omp = op_.ScanRegistration(o), op_.LaserOdometry(o), op_.LaserMapping(o)
    omp.set_frozen(cm, sm)
    omp.set_transform("aft", np.array([0, 0, 0, *start], np.float32))
    out = []
    for t, (pts, rs) in enumerate(sweeps):
        ood.set_features(osr.process(pts, rs))
        ood.process()
        if t > 0:
            omp.set_transform("sum", ood.transform_sum)
            omp.register_frozen(ood.last_corner(), ood.last_surf(), omp.associate())
        out.append((np.array(ood.transform_sum), np.array(omp.transform("aft")), ood.stats()["iterations"], omp.stats()["iterations"]))
    return s, out


def test_bench_scale_8_streams_vs_oracle():
    """The BENCHMARKED shape (bench.py: 8 HDL-64E streams against the 1,000,000-point frozen map, look-ahead on, the bench's own
    trajectories and seeds) for 11 sweeps per stream: every stream against its own oracle chain — accumulated odometry and mapped pose
    within POSE_TOL after every sweep, odometry and mapping iteration counts equal."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    from loam_velodyne_amd import dist as lxdist
    NS, T = 8, 11
    world = synth.World(half_extent=125.0)
    cm, sm = world.make_map(1000000)
    jobs, starts = [], []
    for s, gs in enumerate(lxdist.stream_ids(0, 1, NS)):
        start = lxdist.stream_start(gs)
        starts.append(start)
        poses = synth.trajectory(T, start=start)
        jobs += [(125.0, "HDL-64E", poses[t], poses[t + 1], 1000 * gs + t) for t in range(T)]
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count() or 1), mp_context=mp.get_context("spawn")) as ex:
        made = list(ex.map(synth.make_sweep_job, jobs, chunksize=4))
        sweeps = [[made[s * T + t] for t in range(T)] for s in range(NS)]
        chains = dict(ex.map(_bench_scale_chain, [(s, starts[s], cm, sm, sweeps[s]) for s in range(NS)]))
    pipe = loamx.Pipeline(NS)
    pipe.set_frozen(cm, sm)
    for s in range(NS):
        pipe.set_state(s, aft=np.array([0, 0, 0, *starts[s]], np.float32))
    pipe.upload([[sweeps[s][t] for s in range(NS)] for t in range(T)])
    got = []
    for t in range(T):
        assert pipe.step(t) == (loamx.SKIPPED if t == 0 else loamx.OK)
        got.append([pipe.get(s) for s in range(NS)])
    pipe.close()
    worst = 0.0
    for s in range(NS):
        for t in range(T):
            ts_o, aft_o, oi, mi = chains[s][t]
            tr, ts, aft, st = got[t][s]
            worst = max(worst, np.abs(ts - ts_o).max(), np.abs(aft - aft_o).max())
            assert np.abs(ts - ts_o).max() < POSE_TOL and np.abs(aft - aft_o).max() < POSE_TOL, (s, t)
            if t > 0:
                assert st["odom_iterations"] == oi and st["map_iterations"] == mi, (s, t)
    print("bench-scale parity: worst |difference| %.2e over %d streams x %d sweeps" % (worst, NS, T))


@pytest.mark.parametrize("ahead,pinned", [(3, False), (8, False), (7, True)])
def test_streaming_io_equals_staged_run(orc, small_world, ahead, pinned):
    """(pinned: the destinations are pinned memory of the runtime's own, the downloads go to the SDMA engine directly — csrc/hostlink.hpp;
    ahead = 8: the header's contract to the letter — eight steps in flight, stage_step(t) right after step(t - 8))
    loamx_pipeline_stage_step / download_step_async (the PCIe-inclusive mode): one step handed over at a time, at most eight
    in flight, registered clouds copied out asynchronously from alternating device buffers — bit-identical to the run that
    staged everything up front, and the downloaded clouds are the ones download_full_res returns"""
    ns, T = 2, 12
    cm, sm = small_world.make_map(60000)
    sweeps, starts = [[None] * ns for _ in range(T)], []
    for s in range(ns):
        poses = synth.trajectory(T, start=(1.5 * s, 0.0, 2.0 * s))
        starts.append(np.array([0, 0, 0, 1.5 * s, 0, 2.0 * s], np.float32))
        for t in range(T):
            sw = synth.make_sweep(small_world, "VLP-16", poses[t], poses[t + 1], seed=30 * s + t, az_steps=900)
            sweeps[t][s] = (np.ascontiguousarray(sw.points, np.float32), sw.ring_sizes)

    def make():
        p = loamx.Pipeline(ns)
        p.set_frozen(cm, sm)
        for s in range(ns):
            p.set_state(s, aft=starts[s])
        return p
    a = make()
    a.upload(sweeps)
    ref, ref_full = [], []
    for t in range(T):
        rc = a.step(t)
        ref.append([a.get(s) for s in range(ns)])
        ref_full.append([a.download_full_res(k, len(sweeps[t][k][0])) for k in range(ns)] if rc == loamx.OK else None)
    b = make()
    b.enable_async_downloads()
    if pinned:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")

        def pinned_array(rows):
            ptr = C.c_void_p()
            assert hip.hipHostMalloc(C.byref(ptr), C.c_size_t(rows * 16), C.c_uint(0)) == 0
            return np.ctypeslib.as_array((C.c_float * (rows * 4)).from_address(ptr.value)).reshape(rows, 4)
        # ... and the sweeps come from pinned memory of the runtime's own, one block per step: the staging copies go through ROCr too
        staged = []
        for t in range(T):
            blk = pinned_array(sum(len(sweeps[t][s][0]) for s in range(ns)))
            row, views = 0, []
            for s in range(ns):
                n = len(sweeps[t][s][0])
                blk[row:row + n] = sweeps[t][s][0]
                views.append((blk[row:row + n], sweeps[t][s][1]))
                row += n
            staged.append(views)
        sweeps = staged
    for t in range(min(ahead, T)):
        b.stage_step(t, sweeps[t])
    if pinned:
        outs = [[pinned_array(len(sweeps[0][k][0]) + 8) for k in range(ns)] for _ in range(2)]   # (left to the process' exit)
    else:
        outs = [[np.zeros((len(sweeps[0][k][0]) + 8, 4), np.float32) for k in range(ns)] for _ in range(2)]
    pending = None
    for t in range(T):
        rc = b.step(t)
        if t + ahead < T:
            b.stage_step(t + ahead, sweeps[t + ahead])               # ahead = 7: slot (t + 7) % 8, free since step t - 1 has run; 8: step t's own slot
        for s in range(ns):
            got, want = b.get(s), ref[t][s]
            for i in range(3):
                assert np.array_equal(got[i], want[i]), (t, s, i)
            assert got[3] == want[3]
        if pending is not None:                                       # the previous step's clouds have had a whole step to land
            b.wait_downloads()
            pt, counts, bufs = pending
            for k in range(ns):
                assert counts[k] == len(ref_full[pt][k]) and np.array_equal(bufs[k][:counts[k]], ref_full[pt][k]), (pt, k)
            pending = None
        if rc == loamx.OK:
            bufs = outs[t & 1]
            pending = (t, b.download_step_async(bufs), bufs)
    b.wait_downloads()
    pt, counts, bufs = pending
    for k in range(ns):
        assert np.array_equal(bufs[k][:counts[k]], ref_full[pt][k])
    direct, via_hip = b.download_counts()
    assert (direct == T - 1 and via_hip == 0) if pinned else (direct == 0 and via_hip == T - 1)
    with pytest.raises(loamx.LoamxError):                            # out of order
        b.stage_step(T + 3, sweeps[0])


def _raw_run(small_world, T=6, with_imu=False, sensor="VLP-16", az=900):
    """two streams of raw VLP-16 revolutions (a few NaN / zero returns), optional IMU messages for stream 0"""
    ns = 2
    raws, times = [[None] * ns for _ in range(T)], [[0.0] * ns for _ in range(T)]
    starts = []
    for s in range(ns):
        poses = synth.trajectory(T, start=(1.5 * s, 0.0, 2.0 * s))
        starts.append(np.array([0, 0, 0, 1.5 * s, 0, 2.0 * s], np.float32))
        for t in range(T):
            sw = synth.make_sweep(small_world, sensor, poses[t], poses[t + 1], seed=60 * s + t, az_steps=az)
            raws[t][s] = synth.to_raw(sw, bad_every=41 + s)
            times[t][s] = 0.1 * (t + 1)
    imu_msgs = []
    if with_imu:
        t_imu = 0.0
        while t_imu < 0.1 * (T + 1) + 0.05:
            imu_msgs.append((t_imu, 0.004 * np.sin(3 * t_imu), 0.003 * np.cos(2 * t_imu), 0.05 * t_imu, (0.3 * np.sin(5 * t_imu), 0.05, -0.2 * np.cos(4 * t_imu))))
            t_imu += 0.0031
    return raws, times, starts, imu_msgs


@pytest.mark.parametrize("sensor,az", [("VLP-16", 900), ("HDL-32", 700), ("HDL-64E", 512)])
def test_raw_sweeps_into_the_pipeline_equal_binned_rings(orc, small_world, sensor, az):
    """loamx_pipeline_stage_step_raw without IMU data: the step consumes /velodyne_points payloads; bit-identical to the run that is
    fed the rings the single-sweep entry point bins (loamx_scanreg_process_raw) — i.e. to MultiScanRegistration.cpp:160-238 + the rest"""
    T, ns = 5, 2
    raws, _, starts, _ = _raw_run(small_world, T, sensor=sensor, az=az)
    cm, sm = small_world.make_map(60000)

    def make():
        p = loamx.Pipeline(ns)
        p.set_frozen(cm, sm)
        for s in range(ns):
            p.set_state(s, aft=starts[s])
        return p
    sr = loamx.ScanRegistration()
    binned = [[None] * ns for _ in range(T)]
    for t in range(T):
        for s in range(ns):
            g = sr.process_raw(raws[t][s], sensor)
            binned[t][s] = (g["full"], g["ring_sizes"])
            if t == 0:                                   # ... whose binning is the oracle's (= the reference's, tests/test_ref_pinning.py)
                o_pts, o_rs = op.multiscan_bin(orc, raws[t][s], sensor)
                assert np.array_equal(g["ring_sizes"], o_rs) and np.array_equal(g["full"][:, :3], o_pts[:, :3])
    a = make()
    a.upload(binned)
    b = make()
    for t in range(3):
        b.stage_step_raw(t, raws[t], sensor)
    for t in range(T):
        ra, rb = a.step(t), b.step(t)
        assert ra == rb
        if t + 3 < T:
            b.stage_step_raw(t + 3, raws[t + 3], sensor)
        for s in range(ns):
            ga, gb = a.get(s), b.get(s)
            for i in range(3):
                assert np.array_equal(ga[i], gb[i]), (t, s, i)
            assert ga[3] == gb[3]


def test_raw_sweeps_with_imu_feeds(orc, small_world):
    """per-stream IMU feeds (loamx_pipeline_update_imu): stream 0 gets IMU messages, stream 1 none.  The de-skewed sweeps, the
    imuTransform plugged into the odometry and the registration follow the oracle chain (ScanRegistration with IMU ->
    LaserOdometry.updateIMU -> frozen-map registration WITH transformUpdate's mapping-side roll / pitch blend,
    BasicLaserMapping.cpp:171-200: the same messages feed LaserMapping's history on both sides) within the pose bar; stream 1 is
    untouched by its neighbour's IMU."""
    T, ns = 6, 2
    raws, times, starts, imu_msgs = _raw_run(small_world, T, with_imu=True)
    cm, sm = small_world.make_map(60000)
    p = loamx.Pipeline(ns)
    p.set_frozen(cm, sm)
    q = loamx.Pipeline(ns)                      # the same run without any IMU data
    q.set_frozen(cm, sm)
    for s in range(ns):
        p.set_state(s, aft=starts[s])
        q.set_state(s, aft=starts[s])
    osr, ood, omp = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc)
    omp.set_frozen(cm, sm)
    omp.set_transform("aft", starts[0])
    # every IMU message is on both sides before the first sweep (the order in which messages and sweeps interleave changes what
    # reset() can interpolate — ROS timing, not arithmetic — so the comparison fixes it; the 200-deep history wraps on both sides alike)
    for m in imu_msgs:
        p.update_imu(0, *m)
        osr.update_imu(*m)
        omp.update_imu(m[0], m[1], m[2])   # LaserMapping's own subscription to the same topic: IMUState2 {stamp, roll, pitch}
    for t in range(3):
        p.stage_step_raw(t, raws[t], "VLP-16", scan_times=times[t])
        q.stage_step_raw(t, raws[t], "VLP-16", scan_times=times[t])
    worst = 0.0
    blended_any = False
    for t in range(T):
        p.step(t)
        q.step(t)
        if t + 3 < T:
            p.stage_step_raw(t + 3, raws[t + 3], "VLP-16", scan_times=times[t + 3])
            q.stage_step_raw(t + 3, raws[t + 3], "VLP-16", scan_times=times[t + 3])
        o = osr.process_raw(raws[t][0], times[t][0], "VLP-16")
        ood.update_imu(o["imu_trans"])
        ood.set_features(o)
        ood.process()
        if t > 0:
            omp.set_transform("sum", ood.transform_sum)
            omp.set_time(times[t][0])          # laserOdometryTime of transformUpdate's blend (BasicLaserMapping.cpp:171-200)
            guess = omp.associate()
            pose = omp.register_frozen(ood.last_corner(), ood.last_surf(), guess)
            omp.set_transform("bef", ood.transform_sum)
            omp.set_transform("aft", pose)
            blended_any = blended_any or bool(omp.stats()["optimized"])
        g0, g1, h1 = p.get(0), p.get(1), q.get(1)
        worst = max(worst, float(np.abs(g0[1] - ood.transform_sum).max()))
        assert np.abs(g0[1] - ood.transform_sum).max() < POSE_TOL, (t, g0[1], ood.transform_sum)
        if t > 0:
            assert np.abs(g0[2] - omp.transform("aft")).max() < POSE_TOL, t
        for i in range(3):                                                         # the stream without IMU data is bit-identical
            assert np.array_equal(g1[i], h1[i]), (t, i)
    assert not np.array_equal(p.get(0)[1], q.get(0)[1])                            # ... and the IMU really acted on stream 0
    assert blended_any   # (the mapping-side blend ran on the oracle's side: the mapped poses above include it)
    # ... and the comparison can tell: the blend moves rot_x / rot_z by 0.2 % of (IMU angle - optimum) per sweep, a few 1e-6 rad here,
    # so the blended run sits within 2e-6 of the oracle in the angles while a run with the blend switched off is visibly further away
    if loamx.build_info().get("diag") != "1":   # (LOAMX_NO_MAP_IMU_BLEND changes results: only a `make EXTRA=-DLOAMX_DIAG` build reads it)
        assert np.abs(p.get(0)[2][[0, 2]] - omp.transform("aft")[[0, 2]]).max() < 5e-7
        print("worst odometry difference vs the oracle chain with IMU:", worst, "(contrast run without the blend: diagnostic builds only)")
        return
    import subprocess, sys, json
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_pipeline as tp; from loam_velodyne_amd import loamx, synth;"
            "w = synth.World(half_extent=45.0); raws, times, starts, msgs = tp._raw_run(w, %d, with_imu=True); cm, sm = w.make_map(60000);"
            "p = loamx.Pipeline(2); p.set_frozen(cm, sm); [p.set_state(s, aft=starts[s]) for s in range(2)]; [p.update_imu(0, *m) for m in msgs];"
            "[p.stage_step_raw(t, raws[t], 'VLP-16', scan_times=times[t]) for t in range(3)];"
            "[(p.step(t), (t + 3 < %d) and p.stage_step_raw(t + 3, raws[t + 3], 'VLP-16', scan_times=times[t + 3])) for t in range(%d)];"
            "print(json.dumps([float(x) for x in p.get(0)[2]]))") % (ROOT, os.path.join(ROOT, "tests"), T, T, T)
    env = dict(os.environ, LOAMX_NO_MAP_IMU_BLEND="1", PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    unblended = np.array(json.loads(out.stdout.strip().splitlines()[-1]), np.float32)
    blended, truth = p.get(0)[2], omp.transform("aft")
    assert np.abs(blended[[0, 2]] - truth[[0, 2]]).max() < 5e-7, (blended, truth)
    assert np.abs(unblended[[0, 2]] - truth[[0, 2]]).max() > max(1e-6, 20 * np.abs(blended[[0, 2]] - truth[[0, 2]]).max()), (unblended, blended, truth)
    print("worst odometry difference vs the oracle chain with IMU:", worst)



def test_map_epoch_with_merge_step(orc, small_world):
    """A whole map epoch, end to end (collective 3): the streams register N sweeps against a FROZEN map (epoch 0); their
    sweeps are merged into the map on the side (loamx_pipeline_download_last_clouds + the stream's transformAftMapped ->
    loamx_map_insert, the reference's own insertion + per-cube re-filtering, BasicLaserMapping.cpp:512-593); the merged map is staged,
    indexed in the background and swapped in (epoch 1); the streams go on against it.
      * epoch 0 is bit-identical to a run that never heard of epochs;
      * the merged map holds every point the old one held (up to the re-filtering inside a voxel) plus the sweeps' points (the insertion
        itself is pinned against the oracle's live mapping in tests/test_gpu_mapping.py::test_epoch_merge_insert_...);
      * epoch 1's poses stay on the ground truth (the merged map is a valid map: same bound as epoch 0's)."""
    ns, T, E0 = 2, 8, 4
    cm, sm = small_world.make_map(60000)
    data, gts, starts = [], [], []
    for s in range(ns):
        poses = synth.trajectory(T, start=(1.5 * s, 0.0, 2.0 * s))
        gts.append(poses)
        starts.append(np.array([0, 0, 0, 1.5 * s, 0, 2.0 * s], np.float32))
        data.append([synth.make_sweep(small_world, "VLP-16", poses[t], poses[t + 1], seed=70 * s + t, az_steps=900) for t in range(T)])
    batches = [[(data[s][t].points, data[s][t].ring_sizes) for s in range(ns)] for t in range(T)]

    def make():
        p = loamx.Pipeline(ns)
        p.set_frozen(cm, sm)
        for s in range(ns):
            p.set_state(s, aft=starts[s])
        p.upload(batches)
        return p
    plain = make()
    ref = []
    for t in range(T):
        plain.step(t)
        ref.append([plain.get(s) for s in range(ns)])
    plain.close()

    p = make()
    acc = loamx.LaserMapping()                      # the epoch's accumulator starts from the frozen map
    acc.load_cubes(cm, sm)
    n0 = len(acc.cubes("corner")) + len(acc.cubes("surf"))
    from loam_velodyne_amd import dist as lxdist
    merged = 0
    for t in range(E0):
        p.step(t)
        mine = []
        for s in range(ns):
            got = p.get(s)
            for i in range(3):
                assert np.array_equal(got[i], ref[t][s][i]), (t, s, i)
            if not got[3]["mapped"]:
                continue
            lc, ls = p.last_clouds(s, len(data[s][t].points))
            assert len(lc) > 50 and len(ls) > 200
            mine.append((got[2], lc, ls))
        # the sweeps reach the accumulator as ONE packed message per rank (loamx_dist_pack_clouds -> unpack -> loamx_map_insert): the
        # path the ranks' clouds take over RCCL (loamx_dist_gatherv), here with a single rank
        merged += lxdist.epoch_merge(mine, acc)
    assert merged == ns * (E0 - 1)
    new_c, new_s = acc.cubes("corner"), acc.cubes("surf")
    assert len(new_c) + len(new_s) > n0
    # the old map's points are still there (re-filtering moves a centroid only inside its voxel: < a voxel diagonal)
    from scipy.spatial import cKDTree
    for old, new, leaf in ((cm, new_c, 0.2), (sm, new_s, 0.4)):
        d, _ = cKDTree(new[:, :3]).query(np.asarray(old)[:, :3])
        assert d.max() < leaf * 1.8, d.max()
    # epoch 1: stage the merged map (indexed on the side), swap, go on
    p.stage_frozen(new_c, new_s)
    assert p.swap_frozen()
    worst = 0.0
    for t in range(E0, T):
        assert p.step(t) == loamx.OK
        for s in range(ns):
            _, _, aft, st = p.get(s)
            assert st["mapped"] and 1 <= st["map_iterations"] <= 10
            err = np.abs(aft[3:] - gts[s][t + 1][3:]).max()
            worst = max(worst, err)
    ref_err = max(np.abs(ref[t][s][2][3:] - gts[s][t + 1][3:]).max() for t in range(E0, T) for s in range(ns))
    assert worst < max(2 * ref_err, 0.05), (worst, ref_err)
    p.close()
    acc.close()
    print(f"map epoch: {merged} sweeps merged, map {n0} -> {len(new_c) + len(new_s)} points; epoch-1 position error {worst:.4f} m (frozen epoch-0 map: {ref_err:.4f} m)")
