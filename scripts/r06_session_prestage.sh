#!/bin/bash
# round 6: the next step's full-resolution re-projection behind this step's first iterations (LOAMX_PRESTAGE=1, -DLOAMX_DIAG), once more
set -u
root=$(pwd); out=$root/gpurun_out/r06_prestage; mkdir -p $out
export TMPDIR=/tmp
export LOAMX_LIB=$root/build/diag/loam/libloamx.so
for r in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export LOAMX_PRESTAGE=1; else unset LOAMX_PRESTAGE; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 9 --long-steps 0 > $out/p${v}_$r.json 2> $out/p${v}_$r.err
    python -c "
import json; d=json.load(open('$out/p${v}_$r.json')); print('prestage $v r$r value %8.0f median %8.0f min %8.0f max %8.0f stage %s' % (d['value'], d['value_median'], d['value_min'], d['value_max'], d['config'].get('stage_ms_per_step')))"
  done
done
