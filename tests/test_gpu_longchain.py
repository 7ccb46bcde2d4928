"""GPU: a free-running chain of 100 HDL-64E sweeps against the 1 M-point frozen map (the metric's configuration, one stream) — the parity
rule of the bench line's long window asserted as a test:

  * per step from identical state — the device registration on the oracle chain's own inputs of every sweep — within 1e-4, flat;
  * free running — both chains feed their own poses forward — within max(1e-4, the reference's own envelope over the same sweeps): the
    largest mapped-pose difference the reference's code shows against itself between builds that differ only in what it does not pin
    (FMA contraction under its README's -march=native; the accumulation order inside the forwarded Eigen operations); a sweep beyond that
    only as an explained threshold flip (identical-state difference <= 1e-6 there and a discrete count differing);
  * the reference's own translation units reproduce the oracle chain bit for bit over the whole chain (the pin, at length).

The helpers are bench.py's own (pose_error, reference_envelope, per_step_check, EnvelopeJobs): what the test asserts is what the line gates on."""
import importlib
import multiprocessing as mp
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu


def test_hdl64_chain_of_100_sweeps_within_the_reference_envelope():
    from loam_velodyne_amd import loamx, synth, dist as lxdist
    import oracle_py as op
    bench = importlib.import_module("bench")
    N, LOOK, M = 100, 6, 1_000_000
    T = 1 + N
    world = synth.World(half_extent=125.0)
    cm, sm = world.make_map(M)
    g0 = lxdist.stream_start(0)
    start = np.array([0, 0, 0, g0[0], g0[1], g0[2]], np.float32)
    poses = synth.trajectory(T + LOOK, yaw_step_deg=1.43, start=g0)
    jobs = [(125.0, "HDL-64E", poses[t], poses[t + 1], t) for t in range(T + LOOK)]
    nw = max(1, min(32, len(os.sched_getaffinity(0)), len(jobs)))
    with ProcessPoolExecutor(max_workers=nw, mp_context=mp.get_context("spawn")) as ex:
        made = list(ex.map(synth.make_sweep_job, jobs, chunksize=max(1, len(jobs) // (4 * nw))))
    sweeps = [[m_] for m_ in made]
    kinds = ["oracle_fast"] + (["ref", "ref_map_alt", "ref_odom_alt", "ref_both_alt"] if op.RefLaserMapping.available() and op.RefLaserOdometryAlt.available() else [])
    env_jobs = bench.EnvelopeJobs("test", "frozen", np.stack([made[t][0] for t in range(T)]), made[0][1], cm, sm, start, T, kinds)
    try:
        # the device chain: one stream through the pipeline, look-ahead on, as the bench runs it
        p = loamx.Pipeline(1)
        p.set_frozen(cm, sm)
        p.set_state(0, aft=start)
        p.upload(sweeps)
        gpu = []
        for t in range(T):
            p.step(t)
            _, ts, aft, st = p.get(0)
            gpu.append((t, 0, ts.copy(), aft.copy(), st["odom_iterations"], st["map_iterations"], (st["odom_sel"], st["map_sel"], st["corner_ds"], st["surf_ds"])))
        p.close()
        # the oracle chain of record, its inputs, the device on those inputs
        inputs = []
        m = np.concatenate([cm, sm], axis=0)
        chain = bench.oracle_parity_chain(sweeps, [start], m, len(cm), T, inputs=inputs)
        ps = bench.per_step_check(loamx, cm, sm, inputs, chain)
    finally:
        chains = env_jobs.collect()
    assert not chains.get("_errors"), chains.get("_errors")
    orc_rows = np.array([np.concatenate([[c[0]], c[1].astype(np.float64), c[2].astype(np.float64)]) for c in chain])
    env = bench.reference_envelope(chains, orc_rows)
    pe = bench.pose_error(gpu, chain, stream=0, envelope=env, per_step=ps)
    print("free running: mapped max %.2e m (rmse %.2e), bar %.2e; per step from identical state max %.2e; envelope pairs: %s" % (
        pe["mapped_pose"]["max_m"], pe["mapped_pose"]["rmse_m"], pe["bar_free_running"], ps["max_m"],
        {k: "%.2e" % v["mapped_max_m"] for k, v in env["pairs"].items()}))
    assert pe["sweeps"] == N
    assert ps["max_m"] <= 1e-4 and ps["max_rad"] <= 1e-4, ps
    if "ref" in env["pairs"]:   # the oracle IS the reference, over the whole chain
        assert env["pairs"]["ref"]["mapped_max_m"] == 0.0 and env["pairs"]["ref"]["odometry_sum_max_m"] == 0.0
    assert pe["odometry_iterations_equal"] == N and pe["mapping_iterations_equal"] >= N - 2
    assert pe["within_bar"], {k: v for k, v in pe.items() if k not in ("reference_envelope", "bar_rule", "note")}
