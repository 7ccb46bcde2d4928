#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s49; mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_s49
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_s49 -- python $root/bench.py --mode live --steps 30 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/err.txt
kt=$(find /tmp/prof_s49 -name '*kernel_trace.csv' | head -1)
mt=$(find /tmp/prof_s49 -name '*memory_copy_trace.csv' | head -1)
python - "$kt" "$mt" $out/trace_small.csv <<'PY'
import csv, re, sys
w = csv.writer(open(sys.argv[3], 'w'))
rows = list(csv.DictReader(open(sys.argv[1])))
qcol = 'Stream_Id' if 'Stream_Id' in rows[0] and len({r['Stream_Id'] for r in rows}) > 1 else 'Queue_Id'
for r in rows:
    w.writerow([r['Start_Timestamp'], r['End_Timestamp'], re.sub(r'\(.*', '', r['Kernel_Name'])[-36:], r[qcol]])
try:
    for r in csv.DictReader(open(sys.argv[2])):
        w.writerow([r['Start_Timestamp'], r['End_Timestamp'], 'COPY_' + r.get('Direction', '?'), 'copy'])
except Exception as e:
    print('no copy trace', e)
PY
