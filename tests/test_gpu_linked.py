"""GPU: the linked entry points (loamx_scanreg_process_linked -> loamx_odom_process_linked -> loamx_map_process_linked: a sweep handed
from node to node in HBM) against the host-message entry points on the same sweeps — the data flow is the same, so everything the
two chains produce must be equal bit for bit: odometry transforms, the clouds handed on, mapped poses, the registered cloud, the map."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_py as op  # noqa: E402

from loam_velodyne_amd import loamx, synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _chains(world, sensor, n, map_points, az_steps=None):
    cm, sm = world.make_map(map_points)
    poses = synth.trajectory(n)
    kw = {} if az_steps is None else {"az_steps": az_steps}
    sweeps = [synth.make_sweep(world, sensor, poses[t], poses[t + 1], seed=900 + t, **kw) for t in range(n)]
    return cm, sm, sweeps


@pytest.mark.parametrize("sensor,map_points", [("VLP-16", 60_000), ("HDL-32", 150_000)])
def test_linked_chain_equals_host_message_chain(sensor, map_points):
    world = synth.World(half_extent=65.0)
    n = 9   # (the surround cloud is due on the 1st, 6th, ... processed frame: both branches of process() are covered)
    cm, sm, sweeps = _chains(world, sensor, n, map_points)
    sr_a, od_a, mp_a = loamx.ScanRegistration(), loamx.LaserOdometry(), loamx.LaserMapping()
    sr_b, od_b, mp_b = loamx.ScanRegistration(), loamx.LaserOdometry(), loamx.LaserMapping()
    mp_a.load_cubes(cm, sm)
    mp_b.load_cubes(cm, sm)
    landing = np.zeros((max(len(s.points) for s in sweeps), 4), np.float32)
    # the linked chain also meets the ORACLE directly (it used to be compared with the product's own host-message chain
    # only): the same sweeps, free running over its own live map, through the CPU restatement of the reference
    orc = op.Oracle()
    osr, ood, omp = op.ScanRegistration(orc), op.LaserOdometry(orc), op.LaserMapping(orc)
    omp.load_cubes(cm, sm)
    worst_odom = worst_map = 0.0
    for t, sw in enumerate(sweeps):
        ood.set_features(osr.process(sw.points, sw.ring_sizes))
        ood.process()
        omp.set_inputs(ood.last_corner(), ood.last_surf(), ood.full_to_end(), ood.transform_sum)
        omp.process()
        # host messages
        f = sr_a.process(sw.points.copy(), sw.ring_sizes)
        rc_a = od_a.process(f)
        lc, ls = od_a.last_clouds()
        full = od_a.transform_to_end(f["full"])
        mp_a.update_odometry(od_a.transform_sum)
        rcm_a, reg_a = mp_a.process(lc, ls, full)
        # linked
        sr_b.process_linked(sw.points.copy(), sw.ring_sizes)
        rc_b = od_b.process_linked(sr_b)
        rcm_b, reg_b = mp_b.process_linked(od_b, landing)
        assert rc_a == rc_b and rcm_a == rcm_b, (t, rc_a, rc_b, rcm_a, rcm_b)
        assert rc_b == (loamx.SKIPPED if t == 0 else loamx.OK)
        assert np.array_equal(od_a.transform, od_b.transform), t
        assert np.array_equal(od_a.transform_sum, od_b.transform_sum), t
        assert od_a.stats() == od_b.stats(), t
        lc_b, ls_b = od_b.last_clouds()   # (the host getters keep working behind a linked call)
        assert np.array_equal(lc, lc_b) and np.array_equal(ls, ls_b), t
        for which in ("aft", "bef", "tobe", "sum"):
            assert np.array_equal(mp_a.transform(which), mp_b.transform(which)), (t, which)
        assert mp_a.stats() == mp_b.stats(), t
        assert reg_a.shape == reg_b.shape and np.array_equal(reg_a, reg_b), t
        assert mp_a.has_fresh_map() == mp_b.has_fresh_map(), t
        if mp_a.has_fresh_map():
            assert np.array_equal(mp_a.surround(), mp_b.surround()), t
        # against the oracle: the odometry's accumulated transform within 1e-4, the mapped pose within the free-running bound of a LIVE map
        # (2e-3, tests/test_gpu_mapping.py: both chains insert their own registered sweeps, so a difference feeds back through the map)
        d_odom = float(np.abs(np.asarray(od_b.transform_sum) - ood.transform_sum).max())
        d_map = float(np.abs(mp_b.transform("aft") - omp.transform("aft")).max())
        worst_odom, worst_map = max(worst_odom, d_odom), max(worst_map, d_map)
        assert d_odom < 1e-4 and d_map < 2e-3, (t, d_odom, d_map)
        assert od_b.stats()["iterations"] == ood.stats()["iterations"], t
    print(f"linked chain vs oracle ({sensor}): accumulated odometry max {worst_odom:.2e}, mapped pose max {worst_map:.2e}")
    for which in (0, 1):
        assert np.array_equal(mp_a.cubes(which), mp_b.cubes(which)), which
    assert mp_a.stats()["iterations"] >= 1


def test_linked_chain_without_registered_cloud_and_errors():
    world = synth.World(half_extent=65.0)
    cm, sm, sweeps = _chains(world, "VLP-16", 3, 40_000)
    sr, od, mp = loamx.ScanRegistration(), loamx.LaserOdometry(), loamx.LaserMapping()
    mp.load_cubes(cm, sm)
    # nothing handed on yet
    with pytest.raises(loamx.LoamxError) as e:
        mp.process_linked(od)
    assert e.value.code == loamx.E_INVALID
    ref_sr, ref_od, ref_mp = loamx.ScanRegistration(), loamx.LaserOdometry(), loamx.LaserMapping()
    ref_mp.load_cubes(cm, sm)
    for sw in sweeps:
        sr.process_linked(sw.points, sw.ring_sizes)
        od.process_linked(sr)
        od.link_wait()
        rc, reg = mp.process_linked(od)      # no landing area: the registered cloud is not asked for
        assert reg is None
        f = ref_sr.process(sw.points.copy(), sw.ring_sizes)
        ref_od.process(f)
        lc, ls = ref_od.last_clouds()
        ref_mp.update_odometry(ref_od.transform_sum)
        ref_mp.process(lc, ls, ref_od.transform_to_end(f["full"]))
        assert np.array_equal(mp.transform("aft"), ref_mp.transform("aft"))
    # a landing area that is too small is reported, not overrun
    sr.process_linked(sweeps[0].points, sweeps[0].ring_sizes)
    od.process_linked(sr)
    small = np.zeros((16, 4), np.float32)
    with pytest.raises(loamx.LoamxError):
        mp.process_linked(od, small)
    # non-finite input: told at the node that waits for the extraction
    bad = sweeps[1].points.copy()
    bad[len(bad) // 2, 1] = np.nan
    sr.process_linked(bad, sweeps[1].ring_sizes)
    with pytest.raises(loamx.LoamxError) as e:
        od.process_linked(sr)
    assert e.value.code == loamx.E_INVALID


def test_pinned_caller_memory_is_copied_from_and_to_directly():
    """sweeps and landing areas in memory the runtime has pinned (loamx_host_alloc) take the direct-DMA path of the upload and of
    the registered cloud's download: same results as pageable memory through the library's staging blocks, for both kinds of entry point"""
    world = synth.World(half_extent=65.0)
    cm, sm, sweeps = _chains(world, "VLP-16", 4, 40_000)
    pin = loamx.pinned_copy

    results = []
    for pinned in (False, True):
        sr, od, mp = loamx.ScanRegistration(), loamx.LaserOdometry(), loamx.LaserMapping()
        sr2, od2, mp2 = loamx.ScanRegistration(), loamx.LaserOdometry(), loamx.LaserMapping()
        mp.load_cubes(cm, sm)
        mp2.load_cubes(cm, sm)
        n_max = max(len(s.points) for s in sweeps)
        landing = pin(np.zeros((n_max, 4), np.float32)) if pinned else np.zeros((n_max, 4), np.float32)
        out = []
        for sw in sweeps:
            p = pin(sw.points) if pinned else sw.points.copy()
            before = p.copy()
            sr.process_linked(p, sw.ring_sizes)
            od.process_linked(sr)
            rc, reg = mp.process_linked(od, landing)
            assert np.array_equal(p, before)            # the caller's sweep is only read
            # host-message entry points with the same kind of memory; the registration in place in a pinned copy of the re-projected cloud
            f = sr2.process(p, sw.ring_sizes)
            od2.process(f)
            lc, ls = od2.last_clouds()
            full = od2.transform_to_end(f["full"])
            full = pin(full) if pinned else full
            mp2.update_odometry(od2.transform_sum)
            rc2, reg2 = mp2.process(lc, ls, full, inplace=True)
            assert rc == rc2 and np.array_equal(reg, reg2)
            out.append((np.array(od.transform_sum), mp.transform("aft"), reg.copy()))
        results.append(out)
    for a, b in zip(*results):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("max_iterations", [25, 4])
def test_late_less_flat_cloud_changes_nothing(monkeypatch, max_iterations):
    """the linked odometry starts its iterations as soon as the sharp / less-sharp / flat clouds are compacted and takes the less-flat
    cloud when the sweep's tail needs it (the per-ring voxel grid runs beside the first launches); LOAMX_LINK_NO_SPLIT=1 waits for the
    whole extraction first, as before: everything either chain produces is equal bit for bit — also when one launch pair is all a
    sweep may use (max_iterations 4: no second correspondence launch in front of the late hand-over)"""
    world = synth.World(half_extent=65.0)
    cm, sm, sweeps = _chains(world, "VLP-16", 7, 60_000)
    chains = []
    for split in (True, False):
        if split:
            monkeypatch.delenv("LOAMX_LINK_NO_SPLIT", raising=False)
        else:
            monkeypatch.setenv("LOAMX_LINK_NO_SPLIT", "1")
        sr, od, mp = loamx.ScanRegistration(), loamx.LaserOdometry(max_iterations=max_iterations), loamx.LaserMapping()
        mp.load_cubes(cm, sm)
        landing = np.zeros((max(len(s.points) for s in sweeps), 4), np.float32)
        out = []
        for sw in sweeps:
            sr.process_linked(sw.points, sw.ring_sizes)
            rc = od.process_linked(sr)
            lc, ls = od.last_clouds()
            rcm, reg = mp.process_linked(od, landing)
            out.append((rc, np.array(od.transform), np.array(od.transform_sum), od.stats(), lc, ls, rcm, mp.transform("aft"), reg.copy(), mp.stats()))
        chains.append((out, mp.cubes(0), mp.cubes(1)))
    (a, ac, asf), (b, bc, bsf) = chains
    for t, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and x[3] == y[3] and x[6] == y[6] and x[9] == y[9], t
        for k in (1, 2, 4, 5, 7, 8):
            assert np.array_equal(x[k], y[k]), (t, k)
    assert len(a[-1][5]) > 100 and a[-1][3]["iterations"] >= 1
    assert np.array_equal(ac, bc) and np.array_equal(asf, bsf)


def test_prepared_partition_is_adopted_and_changes_nothing(monkeypatch):
    """the next sweep's map partition + sub-map index are prepared behind the update for the predicted pose and adopted when the true
    pose's plan is identical: on a smooth trajectory that is nearly every sweep, and a handle that never speculates produces the same
    poses, registered clouds and map bit for bit"""
    world = synth.World(half_extent=65.0)
    cm, sm, sweeps = _chains(world, "VLP-16", 12, 60_000)
    chains = []
    for spec in (True, False):
        if spec:
            monkeypatch.delenv("LOAMX_MAP_NO_SPECULATION", raising=False)
        else:
            monkeypatch.setenv("LOAMX_MAP_NO_SPECULATION", "1")
        sr, od, mp = loamx.ScanRegistration(), loamx.LaserOdometry(), loamx.LaserMapping()
        mp.load_cubes(cm, sm)
        landing = np.zeros((max(len(s.points) for s in sweeps), 4), np.float32)
        out = []
        for t, sw in enumerate(sweeps):
            sr.process_linked(sw.points, sw.ring_sizes)
            od.process_linked(sr)
            rc, reg = mp.process_linked(od, landing)
            out.append((mp.transform("aft"), mp.transform("tobe"), reg.copy(), mp.stats()))
            if t == 5:
                assert len(mp.cubes(0)) > 0      # a getter between two sweeps leaves the prepared partition usable
        hits, misses = mp.speculation()
        if spec:
            assert hits + misses == len(sweeps) - 1 and hits >= len(sweeps) - 4, (hits, misses)
        else:
            assert hits == 0 and misses == 0
        chains.append((out, mp.cubes(0), mp.cubes(1), mp.surround()))
    (a, ac, asf, asur), (b, bc, bsf, bsur) = chains
    for x, y in zip(a, b):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]) and x[3] == y[3]
    assert np.array_equal(ac, bc) and np.array_equal(asf, bsf) and np.array_equal(asur, bsur)


def test_a_missed_prediction_is_redone_and_changes_nothing(monkeypatch):
    """replay recorded odometry messages into two mapping handles, one preparing the next partition ahead and one not, with jumps in
    the odometry pose that the constant-velocity prediction cannot follow (a turn of 0.6 rad, a leap of 60 m that moves the cube window):
    the prepared work is discarded on those sweeps and everything the two handles produce stays equal bit for bit"""
    world = synth.World(half_extent=65.0)
    cm, sm, sweeps = _chains(world, "VLP-16", 10, 60_000)
    sr, od = loamx.ScanRegistration(), loamx.LaserOdometry()
    msgs = []
    for sw in sweeps:
        f = sr.process(sw.points.copy(), sw.ring_sizes)
        od.process(f)
        lc, ls = od.last_clouds()
        msgs.append((lc, ls, od.transform_to_end(f["full"]), np.array(od.transform_sum, np.float32)))
    jump = {4: np.array([0, 0.6, 0, 0, 0, 0], np.float32), 7: np.array([0, 0, 0, 60.0, 0, 0], np.float32)}
    outs = []
    for spec in (True, False):
        if spec:
            monkeypatch.delenv("LOAMX_MAP_NO_SPECULATION", raising=False)
        else:
            monkeypatch.setenv("LOAMX_MAP_NO_SPECULATION", "1")
        mp = loamx.LaserMapping()
        mp.load_cubes(cm, sm)
        offset = np.zeros(6, np.float32)
        out = []
        for t, (lc, ls, full, s6) in enumerate(msgs):
            offset = offset + jump.get(t, 0)
            mp.update_odometry(s6 + offset)
            rc, reg = mp.process(lc, ls, full)
            out.append((rc, mp.transform("aft"), mp.transform("tobe"), reg, mp.stats()))
        hits, misses = mp.speculation()
        if spec:
            assert misses >= 2 and hits >= 4, (hits, misses)
        out.append((mp.cubes(0), mp.cubes(1)))
        outs.append(out)
    a, b = outs
    for x, y in zip(a[:-1], b[:-1]):
        assert x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]) and np.array_equal(x[3], y[3]) and x[4] == y[4]
    assert np.array_equal(a[-1][0], b[-1][0]) and np.array_equal(a[-1][1], b[-1][1])
