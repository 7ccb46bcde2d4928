"""Ad-hoc GPU probe: odometry (libloamx) vs the oracle over a short trajectory."""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from loam_velodyne_amd import synth, loamx
import oracle_py as op
orc = op.Oracle()
sensor = sys.argv[1] if len(sys.argv) > 1 else "VLP-16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
w = synth.World(half_extent=65.0)
poses = synth.trajectory(n)
osr = op.ScanRegistration(orc); ood = op.LaserOdometry(orc); god = loamx.LaserOdometry()
worst = 0
for k in range(n):
    sw = synth.make_sweep(w, sensor, poses[k], poses[k+1], seed=k)
    f = osr.process(sw.points, sw.ring_sizes)
    t0=time.time(); ood.set_features(f); ood.process(); t1=time.time()
    god.process(f); t2=time.time()
    dt = np.abs(ood.transform - god.transform).max(); ds = np.abs(ood.transform_sum - god.transform_sum).max()
    oc, os_ = ood.last_corner(), ood.last_surf(); gc, gs = god.last_clouds()
    dc = np.abs(oc-gc).max() if oc.shape==gc.shape else -1; dsf = np.abs(os_-gs).max() if os_.shape==gs.shape else -1
    fe_o = ood.full_to_end(); fe_g = god.transform_to_end(f['full'])
    print(k, 'transform diff %.2e sum diff %.2e lastC %.2e lastS %.2e fullEnd %.2e' % (dt, ds, dc, dsf, np.abs(fe_o-fe_g).max()),
          ood.stats(), god.stats(), 'oracle ms %.2f gpu ms %.2f' % ((t1-t0)*1e3, (t2-t1)*1e3))
    print('    ', np.round(god.transform, 5), np.round(god.transform_sum, 5))
    worst = max(worst, dt, ds)
print('worst', worst)
