#!/bin/bash
# Kernel + copy trace of several PCIe-inclusive windows in one process (which engine carried the block copies in the slow ones?)
set -u
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_$tag -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 1 --ab-pcie "$1" > $out/bench.json 2> $out/bench.err
grep "ab-pcie" $out/bench.err
kt=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1); mc=$(find /tmp/prof_$tag -name '*memory_copy_trace.csv' | head -1)
python - "$kt" "$mc" <<'PY'
import csv, sys
kt = list(csv.DictReader(open(sys.argv[1]))); mc = list(csv.DictReader(open(sys.argv[2])))
t0 = min(int(r['Start_Timestamp']) for r in kt)
big = []
for r in kt:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'copyBuffer' in r['Kernel_Name'] and int(r['Grid_Size_X']) > 10000 and e - s > 150000: big.append(((s - t0) / 1e6, 'blit', (e - s) / 1e3))
for r in mc:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e - s > 150000: big.append(((s - t0) / 1e6, r['Direction'][12:], (e - s) / 1e3))
big.sort()
# windows = bursts separated by > 30 ms
groups, cur = [], []
for b in big:
    if cur and b[0] - cur[-1][0] > 30: groups.append(cur); cur = []
    cur.append(b)
if cur: groups.append(cur)
for g in groups:
    import collections
    c = collections.Counter(x[1] for x in g)
    d = {k: round(sum(x[2] for x in g if x[1] == k) / c[k], 1) for k in c}
    span = g[-1][0] - g[0][0]
    print(f"t={g[0][0]:9.1f} ms  span {span:7.2f} ms  copies {dict(c)}  mean us {d}")
PY
gzip -c "$kt" > $out/kernel_trace.csv.gz; gzip -c "$mc" > $out/memory_copy_trace.csv.gz
