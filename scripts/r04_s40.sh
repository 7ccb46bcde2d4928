#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s40; mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_s40
LOAMX_BENCH_TIMING_PERIOD=1000 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s40 -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 1 > $out/bench.json 2> $out/err.txt
kt=$(find /tmp/prof_s40 -name '*kernel_trace.csv' | head -1)
python - "$kt" $out/trace_small.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
qcol = 'Stream_Id' if 'Stream_Id' in rows[0] and len({r['Stream_Id'] for r in rows}) > 1 else 'Queue_Id'
w = csv.writer(open(sys.argv[2], 'w'))
for r in rows:
    w.writerow([r['Start_Timestamp'], r['End_Timestamp'], re.sub(r'\(.*', '', r['Kernel_Name'])[-36:], r[qcol], r.get('Grid_Size_X', ''), r.get('Grid_Size_Y', '')])
PY
ls -la $out; tail -c 400 $out/bench.json
