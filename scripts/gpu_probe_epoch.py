"""Host-side cost of the double-buffered map epoch calls (stage / swap) and of the build itself."""
import time, sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loam_velodyne_amd import loamx, synth
dev = torch.device("cuda:0")
w = synth.World()
cm, sm = w.make_map(1000000)
m = torch.from_numpy(np.concatenate([cm, sm]).astype(np.float32)).to(dev)
m2 = torch.empty_like(m)
torch.cuda.synchronize()
b = loamx.Batch(4)
b.set_frozen_device(m.data_ptr(), len(cm), m.data_ptr() + 16 * len(cm), len(sm))
torch.cuda.synchronize()
for it in range(4):
    t0 = time.perf_counter(); m2.copy_(m, non_blocking=True); torch.cuda.current_stream().synchronize(); t1 = time.perf_counter()
    b.stage_frozen_device(m2.data_ptr(), len(cm), m2.data_ptr() + 16 * len(cm), len(sm)); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    ok = b.swap_frozen(); t4 = time.perf_counter()
    print("iter %d: copy+sync %.0f us, stage call %.0f us, build (device sync) %.0f us, swap %.0f us" % (it, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6), ok)
