#!/bin/bash
# round 6: the chains' steady-state waits polling instead of blocking (LOAMX_WAIT_SPIN=1), now that the blocking copies are gone:
# 40 batched windows per process (slowest window, median) and the sequential configurations
set -u
root=$(pwd); out=$root/gpurun_out/r06_spinwait; mkdir -p $out
export TMPDIR=/tmp
for r in 1 2 3; do
  for v in 0 1; do
    export LOAMX_WAIT_SPIN=$v
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 40 --long-steps 0 > $out/b${v}_$r.json 2> $out/b${v}_$r.err
    python -c "
import json; d=json.load(open('$out/b${v}_$r.json')); print('batched spin $v r$r value %8.0f median %8.0f min %8.0f max %8.0f' % (d['value'], d['value_median'], d['value_min'], d['value_max']))"
  done
done
live() {  # name sensor map_points
  timeout 300 python bench.py --mode live --sensor $2 --map-points $3 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/$1.json 2> $out/$1.err
  python -c "
import json; d=json.load(open('$out/$1.json')); print('$1', d['value'], d['config']['stage_ms_per_sweep'])"
}
for r in 1 2 3 4; do
  for v in 0 1; do
    export LOAMX_WAIT_SPIN=$v
    live vlp16_spin${v}_$r VLP-16 200000
    live hdl32_spin${v}_$r HDL-32 500000
  done
done
