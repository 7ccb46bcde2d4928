#!/bin/bash
# round 6: registration stream at middle priority — the other configurations (2 M-point map, 32 streams) before adopting it
set -u
root=$(pwd); out=$root/gpurun_out/r06_prio2; mkdir -p $out
export TMPDIR=/tmp
export LOAMX_LIB=$root/build/diag/loam/libloamx.so
run() {  # name args...
  n=$1; shift
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 5 --long-steps 0 "$@" > $out/$n.json 2> $out/$n.err
  python -c "
import json; d=json.load(open('$out/$n.json')); print('%-22s value %8.0f median %8.0f max %8.0f stage %s' % ('$n', d['value'], d['value_median'], d['value_max'], d['config'].get('stage_ms_per_step')))"
}
for r in 1 2; do
  unset LOAMX_PRIO_REG; run map2m_default_$r --map-points 2000000
  export LOAMX_PRIO_REG=0; run map2m_reg0_$r --map-points 2000000
  unset LOAMX_PRIO_REG; run s32_default_$r --streams 32
  export LOAMX_PRIO_REG=0; run s32_reg0_$r --streams 32
  unset LOAMX_PRIO_REG; run s8_default_$r
  export LOAMX_PRIO_REG=0; run s8_reg0_$r
done
