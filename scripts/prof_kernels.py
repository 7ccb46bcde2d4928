"""Average duration per kernel name over the steady-state part of a rocprofv3 kernel-trace CSV."""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', r['Kernel_Name'])[-44:]) for r in rows)
idx = [i for i, e in enumerate(ev) if 'k_feat_ring' in e[2]]
ev = ev[idx[len(idx) // 2]:]          # second half of the run
nstep = len([e for e in ev if 'k_feat_ring' in e[2]])
d = defaultdict(list)
for e in ev: d[e[2]].append((e[1] - e[0]) / 1e3)
print('steps', nstep)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print('  %-46s per-step %8.1f us  n/step=%5.1f  avg=%6.1f  max=%6.1f' % (k, sum(v) / nstep, len(v) / nstep, sum(v) / len(v), max(v)))
