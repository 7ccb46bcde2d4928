"""Per-queue timeline of one steady-state bench step from a rocprofv3 kernel-trace CSV (which chain is critical)."""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
qcol = 'Stream_Id' if 'Stream_Id' in rows[0] and len({r['Stream_Id'] for r in rows}) > 1 else 'Queue_Id'
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', r['Kernel_Name'])[-40:], r[qcol]) for r in rows)
key = sys.argv[3] if len(sys.argv) > 3 else 'k_feat_ring'
idx = [i for i, e in enumerate(ev) if key in e[2]]
a, b = idx[-4], idx[-3]
t0, t1 = ev[a][0], ev[b][0]
print('columns', qcol, 'step period us %.1f' % ((t1 - t0) / 1e3))
byq = defaultdict(list)
for e in ev:
    if t0 <= e[0] < t1: byq[e[3]].append(e)
for q, L in byq.items():
    busy = sum(e[1] - e[0] for e in L) / 1e3
    print('queue %s: n=%d first %.1f last_end %.1f busy %.1f' % (q, len(L), (L[0][0] - t0) / 1e3, (max(e[1] for e in L) - t0) / 1e3, busy))
    if len(sys.argv) > 2:
        prev = None
        for e in L:
            gap = (e[0] - prev) / 1e3 if prev else 0.0
            print('    %8.1f +%6.1f  gap %5.1f  %s' % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, gap, e[2]))
            prev = e[1]
