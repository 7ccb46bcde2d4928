"""Summary of a LOAMX_PIPE_TRACE=1 run (pipeline.hip: one line per step on stderr): per step of the timed window, how long the
registration waited for the step's odometry (M-start), how long its own work took (M-start -> M-downloaded), what the caller spent
between two steps, and the length of each odometry chain's passes.  usage: pipe_trace_summary.py trace.txt [bench.json]"""
import json
import re
import sys

import numpy as np

rows = []
for line in open(sys.argv[1], errors="replace"):
    m = re.match(r"\[pipe t=(\d+)\] caller gap (\S+) \| M-start (\S+)\s+M-enqueued (\S+)\s+M-downloaded (\S+)\s+O-joined (\S+)", line)
    if not m:
        continue
    t, gap, ms, me, md, oj = (float(x) for x in m.groups())
    passes = [float(x) for x in re.findall(r"pass (\S+) us", line)]
    rows.append((t, gap, ms, me, md, oj, passes))
if not rows:
    print("no trace lines")
    sys.exit(0)
# the timed window = the last run of consecutive steps (the warm-up precedes it in the same process)
K = 40
if len(sys.argv) > 2:
    try:
        K = json.load(open(sys.argv[2]))["steps"]
    except Exception:
        pass
w = rows[-K:]
a = np.array([r[1:6] for r in w])
gap, ms, me, md = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
print("steps in window: %d" % len(w))
print("step period (gap + step) us: mean %.1f  median %.1f" % ((gap + a[:, 4]).mean(), np.median(gap + a[:, 4])))
print("caller between steps us: mean %.1f" % gap.mean())
print("waiting for the odometry (M-start) us: mean %.1f  median %.1f  steps with > 20 us: %d" % (ms.mean(), np.median(ms), int((ms > 20).sum())))
print("registration own time (M-start -> downloaded) us: mean %.1f  median %.1f  p90 %.1f" % ((md - ms).mean(), np.median(md - ms), np.percentile(md - ms, 90)))
print("  of which enqueue (M-start -> M-enqueued): mean %.1f" % (me - ms).mean())
np_ = max(len(r[6]) for r in w)
for c in range(np_):
    p = np.array([r[6][c] for r in w if len(r[6]) > c])
    print("odometry chain %d pass us (most recent at each step): mean %.1f  median %.1f  p90 %.1f  max %.1f" % (c, p.mean(), np.median(p), np.percentile(p, 90), p.max()))
if len(sys.argv) > 2:
    try:
        d = json.load(open(sys.argv[2]))
        print("bench: value %.0f sweeps/s, ms_per_step %.4f" % (d["value"], d["ms_per_step"]))
    except Exception as e:
        print("bench line unreadable:", e)
