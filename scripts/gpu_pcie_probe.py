"""GPU probe: where does the PCIe-inclusive step time go?  Streams the bench workload with (a) H2D only, (b) D2H only, (c) both,
and times the host calls."""
import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loam_velodyne_amd import loamx, synth

ns, T = 8, 14
w = synth.World(half_extent=125.0)
cm, sm = w.make_map(1_000_000)
map_t = torch.from_numpy(np.concatenate([cm, sm])).cuda()
sweeps, starts = [[None] * ns for _ in range(T)], []
for s in range(ns):
    start = (3.0 * s - 10.0, 0.0, -40.0 + 2.0 * s)
    poses = synth.trajectory(T, start=start)
    starts.append(np.array([0, 0, 0, *start], np.float32))
    for t in range(T):
        sw = synth.make_sweep(w, "HDL-64E", poses[t], poses[t + 1], seed=1000 * s + t)
        pts = torch.from_numpy(np.ascontiguousarray(sw.points, np.float32)).pin_memory()
        sweeps[t][s] = (pts.numpy(), sw.ring_sizes, pts)
n_pts = len(sweeps[0][0][0])
outs = [[torch.empty((n_pts + 8, 4), dtype=torch.float32).pin_memory() for _ in range(ns)] for _ in range(2)]


def run(label, h2d, d2h):
    p = loamx.Pipeline(ns)
    p.set_frozen_device(map_t.data_ptr(), len(cm), map_t.data_ptr() + 16 * len(cm), len(sm))
    for k in range(ns):
        p.set_state(k, aft=starts[k])
    if d2h:
        p.enable_async_downloads()
    if h2d:
        for t in range(3):
            p.stage_step(t, [(a, r) for a, r, _ in sweeps[t]])
    else:
        p.upload([[(a, r) for a, r, _ in sweeps[t]] for t in range(T)])
    ts, tstage, tdl = [], [], []
    for t in range(T):
        a = time.perf_counter()
        rc = p.step(t)
        b = time.perf_counter()
        if h2d and t + 3 < T:
            p.stage_step(t + 3, [(x, r) for x, r, _ in sweeps[t + 3]])
        c = time.perf_counter()
        if d2h and rc == loamx.OK:
            p.download_step_async([o.numpy() for o in outs[t & 1]])
        d = time.perf_counter()
        if t >= 4:
            ts.append(b - a); tstage.append(c - b); tdl.append(d - c)
    if d2h:
        p.wait_downloads()
    torch.cuda.synchronize()
    print(f"{label:22s} step {np.median(ts)*1e3:6.3f} ms  stage_step call {np.median(tstage)*1e3:6.3f} ms  download call {np.median(tdl)*1e3:6.3f} ms", flush=True)


run("resident", False, False)
run("H2D streaming", True, False)
run("D2H async", False, True)
run("H2D + D2H", True, True)
# raw copy speed of the same buffers through torch
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dst = torch.empty((ns, n_pts, 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
ev0.record()
for s in range(ns):
    dst[s].copy_(sweeps[0][s][2], non_blocking=True)
ev1.record(); torch.cuda.synchronize()
print("torch H2D of one step's sweeps (16.8 MB, pinned):", ev0.elapsed_time(ev1), "ms")
ev0.record()
for s in range(ns):
    outs[0][s][:n_pts].copy_(dst[s], non_blocking=True)
ev1.record(); torch.cuda.synchronize()
print("torch D2H of one step's clouds (16.8 MB, pinned):", ev0.elapsed_time(ev1), "ms")
