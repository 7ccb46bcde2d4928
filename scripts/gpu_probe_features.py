"""Ad-hoc GPU probe: feature extraction (libloamx) vs the oracle — expected bit-exact."""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from loam_velodyne_amd import synth, loamx
import oracle_py as op
orc = op.Oracle()
w = synth.World(half_extent=65.0)
poses = synth.trajectory(3)
osr = op.ScanRegistration(orc); gsr = loamx.ScanRegistration()
for sensor in ("VLP-16", "HDL-32", "HDL-64E"):
    for k in range(2):
        sw = synth.make_sweep(w, sensor, poses[k], poses[k+1], seed=k)
        t0=time.time(); fo = osr.process(sw.points, sw.ring_sizes); t1=time.time()
        fg = gsr.process(sw.points, sw.ring_sizes); t2=time.time()
        for n in ("sharp","less_sharp","flat","less_flat"):
            same = fo[n].shape == fg[n].shape and np.array_equal(fo[n], fg[n])
            md = np.abs(fo[n]-fg[n]).max() if fo[n].shape == fg[n].shape and len(fo[n]) else -1
            print(sensor, k, n, fo[n].shape, fg[n].shape, 'bit-exact' if same else f'DIFF max {md}')
        print('   oracle ms', (t1-t0)*1e3, 'gpu call ms', (t2-t1)*1e3)
