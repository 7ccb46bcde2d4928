#!/bin/bash
# round 6: stream priorities once more now that the odometry chains bound the step (LOAMX_PRIO_* in a -DLOAMX_DIAG build)
set -u
root=$(pwd); out=$root/gpurun_out/r06_prio; mkdir -p $out
export TMPDIR=/tmp
export LOAMX_LIB=$root/build/diag/loam/libloamx.so
run() {  # name
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 5 --long-steps 0 > $out/$1.json 2> $out/$1.err
  python -c "
import json; d=json.load(open('$out/$1.json')); print('%-22s value %8.0f median %8.0f max %8.0f stage %s' % ('$1', d['value'], d['value_median'], d['value_max'], d['config'].get('stage_ms_per_step')))"
}
for r in 1 2 3; do
  unset LOAMX_PRIO_REG LOAMX_PRIO_FEAT LOAMX_PRIO_ODOM; run default_$r
  export LOAMX_PRIO_REG=0; run reg0_$r; unset LOAMX_PRIO_REG
  export LOAMX_PRIO_REG=-1; run regm1_$r; unset LOAMX_PRIO_REG
  export LOAMX_PRIO_FEAT=0; run feat0_$r; unset LOAMX_PRIO_FEAT
done
