"""Ad-hoc GPU probe: sequential mapping with a live map (libloamx) vs the oracle over a short trajectory.
Features + odometry come from the oracle for both sides so that only BasicLaserMapping is compared."""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from loam_velodyne_amd import synth, loamx
import oracle_py as op
orc = op.Oracle()
sensor = sys.argv[1] if len(sys.argv) > 1 else "VLP-16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w = synth.World(half_extent=65.0)
poses = synth.trajectory(n)
osr = op.ScanRegistration(orc); ood = op.LaserOdometry(orc); omp = op.LaserMapping(orc); gmp = loamx.LaserMapping()
def setdiff(a, b):
    if len(a) != len(b): return 'SIZE %d vs %d' % (len(a), len(b))
    ia = np.lexsort(a[:, :3].T[::-1]); ib = np.lexsort(b[:, :3].T[::-1])
    return '%.2e' % np.abs(a[ia] - b[ib]).max()
worst = 0
for k in range(n):
    sw = synth.make_sweep(w, sensor, poses[k], poses[k+1], seed=k)
    f = osr.process(sw.points, sw.ring_sizes)
    ood.set_features(f); ood.process()
    full_end = ood.full_to_end(); lc, ls, ts = ood.last_corner(), ood.last_surf(), ood.transform_sum
    t0=time.time(); omp.set_inputs(lc, ls, full_end, ts); omp.process(); t1=time.time()
    gmp.update_odometry(ts); rc, gfull = gmp.process(lc, ls, full_end); t2=time.time()
    d = np.abs(omp.transform('aft') - gmp.transform('aft')).max(); db = np.abs(omp.transform('bef') - gmp.transform('bef')).max()
    worst = max(worst, d)
    print(k, 'aft diff %.2e bef diff %.2e full %.2e' % (d, db, np.abs(omp.cloud('full_res') - gfull).max()), omp.stats(), gmp.stats(),
          'oracle ms %.1f gpu ms %.1f' % ((t1-t0)*1e3, (t2-t1)*1e3))
    print('     corner cubes', setdiff(omp.cloud('corner_cubes'), gmp.cubes('corner')), 'surf cubes', setdiff(omp.cloud('surf_cubes'), gmp.cubes('surf')),
          'fresh', omp.has_fresh_map(), gmp.has_fresh_map(), ('surround ' + setdiff(omp.cloud('surround_ds'), gmp.surround())) if gmp.has_fresh_map() else '')
print('worst aft diff', worst)
