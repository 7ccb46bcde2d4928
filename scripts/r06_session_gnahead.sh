#!/bin/bash
# round 6: Gauss-Newton launches enqueued beyond the previous step's need (LOAMX_GN_AHEAD in a -DLOAMX_DIAG build; default 1)
set -u
root=$(pwd); out=$root/gpurun_out/r06_gnahead; mkdir -p $out
export TMPDIR=/tmp
export LOAMX_LIB=$root/build/diag/loam/libloamx.so
for r in 1 2 3; do
  for n in 1 2 3; do
    export LOAMX_GN_AHEAD=$n
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 5 --long-steps 0 > $out/bench_${n}_$r.json 2> $out/err_${n}_$r.txt
    python - <<PY
import json
try:
    d = json.load(open('$out/bench_${n}_$r.json'))
    print('ahead %-2s r$r  value %8.0f  median %8.0f  max %8.0f  stage %s' % ('$n', d['value'], d.get('value_median', 0), d.get('value_max', 0), d['config'].get('stage_ms_per_step')))
except Exception as e:
    print('$n r$r FAILED', e)
PY
  done
done
