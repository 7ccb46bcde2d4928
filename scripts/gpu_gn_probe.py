"""GPU probe: time the Gauss-Newton stage of the batched registration (8 HDL-64E sweeps vs a 1M-pt map) for several
workgroup counts of the persistent kernel and for the round-1 launch-per-iteration loop.  usage: [B] [map_points]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from loam_velodyne_amd import synth, loamx
import oracle_py as op

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
orc = op.Oracle()
w = synth.World(half_extent=125.0)
cm, sm = w.make_map(M)
rng = np.random.default_rng(1)
sr = op.ScanRegistration(orc)
cl, sl, guesses = [], [], []
for k in range(B):
    gt = np.array([0.01 * rng.normal(), 0.3 * rng.normal(), 0.01 * rng.normal(), 3 * rng.normal(), 0.05 * rng.normal(), 3 * rng.normal()])
    sw = synth.make_sweep(w, "HDL-64E", gt, gt, seed=k)
    f = sr.process(sw.points, sw.ring_sizes)
    c, s = f["less_sharp"].copy(), f["less_flat"].copy()
    c[:, 3] = np.floor(c[:, 3]); s[:, 3] = np.floor(s[:, 3])
    cl.append(c); sl.append(s)
    guesses.append(gt + np.array([0.003, 0.003, 0.003, 0.05, 0.05, 0.05]) * rng.normal(size=6))
guesses = np.array(guesses, np.float32)


def run(label, env):
    for k in ("LOAMX_GN_WGS", "LOAMX_REG_LEGACY"):
        os.environ.pop(k, None)
    os.environ.update(env)
    b = loamx.Batch(B)
    b.set_frozen(cm, sm)
    b.upload(cl, sl, guesses)
    b.set_timing(True)
    b.run()
    poses, stats = b.download()
    ts = []
    for _ in range(2 if os.environ.get("PROBE_SHORT") else 6):
        b.upload(cl, sl, guesses)
        b.run()
        ts.append(b.timing())
    t = ts[-1]
    print(f"{label:28s} run {np.median([x['run_ms'] for x in ts]):7.3f} ms  GN {np.median([x['residual_ms'] for x in ts]):7.3f} ms  launches {t['residual_launches']}  "
          f"iters {stats[:, 0].tolist()}  q-iters {t['query_iterations']}", flush=True)
    return poses


run("fused k_gn_iter", {})

# the search routine alone (k_knn_probe) on the same number of queries as one Gauss-Newton iteration of the batch
if os.environ.get("PROBE_KNN"):
    R = [synth.rot_zxy(*g[:3]) for g in guesses]
    qs = np.concatenate([(s_[:, :3].astype(np.float64) @ R[k].T + guesses[k][3:]).astype(np.float32) for k, s_ in enumerate(sl)])
    from scipy.spatial import cKDTree  # only to thin the queries like the voxel grid would (0.4 m)
    keep = np.unique(np.floor(qs / 0.4).astype(np.int64), axis=0, return_index=True)[1]
    qs = qs[np.sort(keep)]
    b = loamx.Batch(1)
    b.set_frozen(cm, sm)
    for _ in range(3):
        t0 = time.perf_counter()
        b.knn_probe(1, qs)
        print(f"knn_probe of {len(qs)} surf queries (incl. copies): {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
