"""Summarise one bench step from a rocprofv3 kernel trace CSV: per-kernel time inside the step, span, gaps."""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', r['Kernel_Name'])[-44:]) for r in rows)
idx = [i for i, e in enumerate(ev) if 'k_feat_point' in e[2]]
a, b = idx[-3], idx[-2]
step = ev[a:b]
t0 = step[0][0]; t1 = max(e[1] for e in step)
print('step span us %.1f  busy %.1f  kernels %d' % ((t1 - t0) / 1e3, sum(e[1] - e[0] for e in step) / 1e3, len(step)))
d = defaultdict(list)
for e in step: d[e[2]].append((e[1] - e[0]) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
    print('  %-46s %8.1f us  n=%-3d each=%s' % (k, sum(v), len(v), ' '.join('%.0f' % x for x in v[:10])))
