"""Raw-sweep ingestion: wall time of loamx_scanreg_process_raw on HDL-64E revolutions (with and
without IMU data) next to the oracle's CPU loop.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import oracle_py as op
from loam_velodyne_amd import loamx, synth

orc = op.Oracle(fast=True) if "fast" in op.Oracle.__init__.__code__.co_varnames else op.Oracle()
w = synth.World()
sw = synth.make_sweep(w, "HDL-64E", np.zeros(6), np.array([0, 0.01, 0, 0.2, 0, 1.0]), seed=1)
raw = synth.to_raw(sw, bad_every=64)
n = len(raw)
for imu in (False, True):
    g, o = loamx.ScanRegistration(), op.ScanRegistration(orc)
    if imu:
        for j in range(200):
            t = 0.001 * j
            for h in (g, o):
                h.update_imu(t, 0.01 * np.sin(t), 0.02 * np.cos(t), 0.3 * t, (0.2, 0.0, -0.1))
    g.set_time(0.1); g.process_raw(raw, "HDL-64E"); o.process_raw(raw, 0.1, "HDL-64E")     # warm-up / first sweep
    t0 = time.perf_counter()
    for k in range(10):
        g.set_time(0.2 + 0.1 * k)
        g.process_raw(raw, "HDL-64E")
    tg = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for k in range(3):
        op.multiscan_bin(orc, raw, "HDL-64E") if not imu else o.process_raw(raw, 0.2 + 0.1 * k, "HDL-64E")
    to = (time.perf_counter() - t0) / 3
    print("imu=%d  points %d  GPU call (H2D + ingestion + features + D2H) %.2f ms   oracle %s %.2f ms" %
          (imu, n, tg * 1e3, "bin+features" if imu else "bin only", to * 1e3))
