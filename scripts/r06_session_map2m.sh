#!/bin/bash
# round 6: the 2 M-point map block's first window came out slow once (11.8 k against 15.9 k): which call? (-DLOAMX_API_TRACE build, then the product)
set -u
root=$(pwd); out=$root/gpurun_out/r06_map2m; mkdir -p $out
export TMPDIR=/tmp
for r in 1 2 3 4 5 6; do
  if [ $r -le 3 ]; then export LOAMX_LIB=$root/build/apitrace/loam/libloamx.so; else unset LOAMX_LIB; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 3 --long-steps 0 --map-points 2000000 > $out/m_$r.json 2> $out/m_$r.err
  python -c "
import json; d=json.load(open('$out/m_$r.json')); print('map2m r$r value %8.0f median %8.0f min %8.0f max %8.0f' % (d['value'], d['value_median'], d['value_min'], d['value_max']))"
  grep -h "api trace" $out/m_$r.err | grep -v "hipStreamCreate" | sort -t' ' -k3 -n -r | head -8
done
