#!/bin/bash
# round 6: host stamps inside Mapper::process (LOAMX_MAP_TRACE) + every device allocation (LOAMX_ALLOC_TRACE) for several live processes:
# where do the slow runs lose their 10 ms?
set -u
root=$(pwd); out=$root/gpurun_out/r06_maptrace; mkdir -p $out
export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  LOAMX_ALLOC_TRACE=1 LOAMX_MAP_TRACE=1 timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/live_$i.json 2> $out/live_$i.err
  python -c "
import json; d=json.load(open('$out/live_$i.json')); print('run $i', d['value'], d['config']['stage_ms_per_sweep']['mapping'])"
  grep -n "mean of\|alloc" $out/live_$i.err | awk -F: '{print $1": "substr($0, index($0,$2), 110)}' | grep -B2 -A2 "slowest call [0-9]\{4,\}" | tail -12
done
