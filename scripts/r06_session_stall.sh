#!/bin/bash
# round 6: where does the occasional ~8 ms call of the sequential-SLAM chain wait? (LOAMX_MAP_TRACE: slow calls + slow helper jobs)
set -u
root=$(pwd); out=$root/gpurun_out/r06_stall; mkdir -p $out
export TMPDIR=/tmp
export LOAMX_MAP_TRACE=1
for r in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  timeout 300 python bench.py --mode live --sensor HDL-32 --map-points 500000 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/hdl32_$r.json 2> $out/hdl32_$r.err
  python -c "
import json; d=json.load(open('$out/hdl32_$r.json')); print('hdl32_$r', d['value'], d['config']['stage_ms_per_sweep'])"
  grep -h "slow call\|slow helper" $out/hdl32_$r.err | grep "slow helper"
done
