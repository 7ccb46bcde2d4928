"""GPU probe: the registration chain of one batch alone on the device (8 HDL-64E sweeps vs a 1M-pt frozen map through
loamx_batch_*), repeated; run it under `rocprofv3 --kernel-trace` for the stand-alone duration of every kernel of the chain
(k_vb_plan / k_vb_stack / k_vb_reduce / k_gn_iter / k_transform_full).  usage: [B] [map_points] [sensor] [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from loam_velodyne_amd import synth, loamx
import oracle_py as op

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
SENSOR = sys.argv[3] if len(sys.argv) > 3 else "HDL-64E"
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 10
orc = op.Oracle()
w = synth.World(half_extent=125.0)
cm, sm = w.make_map(M)
rng = np.random.default_rng(1)
sr = op.ScanRegistration(orc)
cl, sl, guesses, fulls = [], [], [], []
for k in range(B):
    gt = np.array([0.01 * rng.normal(), 0.3 * rng.normal(), 0.01 * rng.normal(), 3 * rng.normal(), 0.05 * rng.normal(), 3 * rng.normal()])
    sw = synth.make_sweep(w, SENSOR, gt, gt, seed=k)
    f = sr.process(sw.points, sw.ring_sizes)
    c, s = f["less_sharp"].copy(), f["less_flat"].copy()
    c[:, 3] = np.floor(c[:, 3]); s[:, 3] = np.floor(s[:, 3])
    cl.append(c); sl.append(s); fulls.append(sw.points.copy())
    guesses.append(gt + np.array([0.003, 0.003, 0.003, 0.05, 0.05, 0.05]) * rng.normal(size=6))
guesses = np.array(guesses, np.float32)
b = loamx.Batch(B)
b.set_frozen(cm, sm)
b.set_timing(True)
ts = []
for r in range(REPS):
    b.upload(cl, sl, guesses, full_res=fulls)
    t0 = time.perf_counter()
    b.run()
    ts.append((time.perf_counter() - t0) * 1e3)
    tm = b.timing()
poses, stats = b.download()
print(f"{SENSOR} B={B} map={M}: run wall ms median {np.median(ts):.3f} min {min(ts):.3f}; device run {tm['run_ms']:.3f} ms, GN launches {tm['residual_launches']} "
      f"({tm['residual_ms']:.3f} ms), iterations {stats[:, 0].tolist()}, queries {tm['queries']}", flush=True)
