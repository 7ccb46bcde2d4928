"""Ad-hoc GPU probe: streaming pipeline (features -> odometry -> frozen-map registration) vs the oracle chain."""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from loam_velodyne_amd import synth, loamx
import oracle_py as op
sensor = sys.argv[1] if len(sys.argv) > 1 else "VLP-16"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
T = int(sys.argv[4]) if len(sys.argv) > 4 else 4
check = (len(sys.argv) <= 5) or sys.argv[5] != "nocheck"
orc = op.Oracle(fast=not check)
w = synth.World(half_extent=65.0 if M <= 300000 else 125.0)
corner_map, surf_map = w.make_map(M)
t0 = time.time()
sweeps = [[None]*NS for _ in range(T)]
for s in range(NS):
    poses = synth.trajectory(T, start=(2.0*s - NS, 0.0, 3.0*s))
    for t in range(T):
        sw = synth.make_sweep(w, sensor, poses[t], poses[t+1], seed=100*s+t)
        sweeps[t][s] = (sw.points, sw.ring_sizes)
    if s == 0: pose0 = poses
print('generated', time.time()-t0)
pipe = loamx.Pipeline(NS)
pipe.set_frozen(corner_map, surf_map)
starts = [np.array([0,0,0, 2.0*s - NS, 0, 3.0*s], np.float32) for s in range(NS)]
for s in range(NS): pipe.set_state(s, aft=starts[s])   # bef = 0 (odometry frame), aft = true start pose in the map
pipe.upload(sweeps)
pipe.set_timing(True)
if check:
    osr = [op.ScanRegistration(orc) for _ in range(NS)]; ood = [op.LaserOdometry(orc) for _ in range(NS)]; omp = [op.LaserMapping(orc) for _ in range(NS)]
    for s in range(NS):
        omp[s].set_frozen(corner_map, surf_map); omp[s].set_transform('aft', starts[s])
worst = 0
for t in range(T):
    t1 = time.time(); rc = pipe.step(t); t2 = time.time()
    print('step', t, 'rc', rc, 'wall ms %.2f' % ((t2-t1)*1e3), pipe.timing())
    for s in range(NS):
        tr, ts, aft, st = pipe.get(s)
        if check:
            f = osr[s].process(*sweeps[t][s]); ood[s].set_features(f); ood[s].process()
            if t > 0:
                omp[s].set_transform('sum', ood[s].transform_sum); g = omp[s].associate()
                omp[s].register_frozen(ood[s].last_corner(), ood[s].last_surf(), g)
            d1 = np.abs(ts - ood[s].transform_sum).max(); d2 = np.abs(aft - omp[s].transform('aft')).max()
            worst = max(worst, d1, d2)
            print('   stream', s, 'sum diff %.2e aft diff %.2e' % (d1, d2), st, np.round(aft, 4))
        elif s == 0:
            print('   stream', s, st, np.round(aft, 4))
print('worst', worst)
