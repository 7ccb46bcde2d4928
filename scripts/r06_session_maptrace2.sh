#!/bin/bash
# round 6: does the helper's spin phase remove the occasional ~10 ms call?  20 live processes, slowest call of each timed block
set -u
root=$(pwd); out=$root/gpurun_out/r06_maptrace2; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_linked.py tests/test_gpu_mapping.py tests/test_gpu_nodes.py -m gpu -x -q 2>&1 | tail -2
for i in $(seq 1 20); do
  LOAMX_MAP_TRACE=1 timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/live_$i.json 2> $out/live_$i.err
  python -c "
import json; d=json.load(open('$out/live_$i.json')); print('run $i', d['value'], d['config']['stage_ms_per_sweep']['mapping'], end=' ')"
  grep "mean of" $out/live_$i.err | sed 's/.*(slowest call \([0-9.]*\)).*/\1/' | tr '\n' ' '; echo
done
