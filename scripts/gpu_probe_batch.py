"""Ad-hoc GPU probe: batched registration (libloamx) vs the oracle on synthetic sweeps.  usage: sensor M B"""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from loam_velodyne_amd import synth, loamx
import oracle_py as op
orc = op.Oracle()
sensor = sys.argv[1] if len(sys.argv) > 1 else "VLP-16"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
w = synth.World(half_extent=65.0 if M <= 300000 else 125.0)
corner_map, surf_map = w.make_map(M)
rng = np.random.default_rng(1)
sr = op.ScanRegistration(orc)
cl, sl, guesses, gts = [], [], [], []
for k in range(B):
    gt = np.array([0.01*rng.normal(), 0.3*rng.normal(), 0.01*rng.normal(), 3*rng.normal(), 0.05*rng.normal(), 3*rng.normal()])
    sw = synth.make_sweep(w, sensor, gt, gt, seed=k)
    f = sr.process(sw.points, sw.ring_sizes)
    c = f['less_sharp'].copy(); s = f['less_flat'].copy()
    c[:,3] = np.floor(c[:,3]); s[:,3] = np.floor(s[:,3])
    cl.append(c); sl.append(s); gts.append(gt)
    guesses.append(gt + np.array([0.003,0.003,0.003,0.05,0.05,0.05])*rng.normal(size=6))
guesses = np.array(guesses, np.float32)
t0 = time.time()
mp = op.LaserMapping(orc)
mp.set_frozen(corner_map, surf_map)
t_build = time.time()-t0
oposes, ostats = [], []
t0 = time.time()
for k in range(B):
    oposes.append(mp.register_frozen(cl[k], sl[k], guesses[k])); ostats.append(mp.stats())
t_orc = time.time()-t0
oposes = np.array(oposes)
b = loamx.Batch(B)
b.set_frozen(corner_map, surf_map)
b.upload(cl, sl, guesses)
b.set_timing(True)
b.run()
gposes, gstats = b.download()
tm = b.timing()
for k in range(B):
    print(k, 'gt   ', np.round(gts[k],5))
    print('   orc ', np.round(oposes[k],5), ostats[k]['iterations'], ostats[k]['sel'], ostats[k]['corner_ds'], ostats[k]['surf_ds'])
    print('   gpu ', np.round(gposes[k],5), gstats[k])
    print('   diff', np.abs(gposes[k]-oposes[k]).max())
print('max abs diff', np.abs(gposes-oposes).max(), 'oracle s/sweep', t_orc/B, 'kd build', t_build, 'gpu', tm)
t0=time.time()
for _ in range(5): b.run()
print('gpu run avg ms', (time.time()-t0)/5*1e3, b.timing())
