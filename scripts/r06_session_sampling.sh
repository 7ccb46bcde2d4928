#!/bin/bash
# round 6: what do the HIP-event-sampled steps cost the timed window?  (LOAMX_BENCH_TIMING_PERIOD: every 4th step (default), every 10th, none)
set -u
root=$(pwd); out=$root/gpurun_out/r06_sampling; mkdir -p $out
export TMPDIR=/tmp
for r in 1 2 3; do
  for p in 4 10 100000; do
    export LOAMX_BENCH_TIMING_PERIOD=$p
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 9 --long-steps 0 > $out/p${p}_$r.json 2> $out/p${p}_$r.err
    python -c "
import json; d=json.load(open('$out/p${p}_$r.json')); print('period %-7s r$r value %8.0f median %8.0f min %8.0f max %8.0f  lm launches %s' % ('$p', d['value'], d['value_median'], d['value_min'], d['value_max'], (d.get('roofline') or {}).get('launches')))"
  done
done
