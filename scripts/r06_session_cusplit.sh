#!/bin/bash
# round 6: compute units in two sets (LOAMX_CU_SPLIT=n in a -DLOAMX_DIAG build: odometry chains on mask bits [0, n), registration +
# features on the rest) — do the chains' iterations run at their stand-alone speed when nothing shares their compute units?
set -u
root=$(pwd); out=$root/gpurun_out/r06_cusplit; mkdir -p $out
export TMPDIR=/tmp
export LOAMX_LIB=$root/build/diag/loam/libloamx.so
for r in 1 2; do
  for n in 0 48 80 112 144; do
    if [ $n = 0 ]; then unset LOAMX_CU_SPLIT; else export LOAMX_CU_SPLIT=$n; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 3 --long-steps 0 > $out/bench_${n}_$r.json 2> $out/err_${n}_$r.txt
    python - <<PY
import json
try:
    d = json.load(open('$out/bench_${n}_$r.json'))
    rf = d.get('roofline', {})
    print('split %-4s r$r  value %8.0f  median %8.0f  max %8.0f  stage %s  lm us/iter %s corr %s' % ('$n', d['value'], d.get('value_median', 0), d.get('value_max', 0), d['config'].get('stage_ms_per_step'), rf.get('latency_model', {}).get('us_per_iteration'), rf.get('corr_pair', {}).get('avg_launch_us')))
except Exception as e:
    print('$n r$r FAILED', e)
PY
  done
done
