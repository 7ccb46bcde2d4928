#!/bin/bash
# round 6: do the batched windows show the milliseconds-long host stalls too?  40 windows of 20 steps per process, the library of the last
# commit against the one whose per-step offset table comes down by a kernel store (build/head: HEAD's sources)
set -u
root=$(pwd); out=$root/gpurun_out/r06_hiccup; mkdir -p $out
export TMPDIR=/tmp
for r in 1 2 3; do
  for v in head new; do
    if [ $v = head ]; then export LOAMX_LIB=$root/build/head/loam/libloamx.so; else unset LOAMX_LIB; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 40 --long-steps 0 > $out/${v}_$r.json 2> $out/${v}_$r.err
    python -c "
import json; d=json.load(open('$out/${v}_$r.json')); print('%-5s r$r value %8.0f median %8.0f min %8.0f max %8.0f' % ('$v', d['value'], d['value_median'], d['value_min'], d['value_max']))"
  done
done
