#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s58; mkdir -p $out
export TMPDIR=/tmp
for v in base lmw3 lmw2 base lmw3; do
  if [ $v = base ]; then unset LOAMX_LIB; else export LOAMX_LIB=build/prof/libloamx_$v.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 > $out/b_$v.json 2> $out/b_$v.err
  python -c "
import json;d=json.load(open('$out/b_$v.json'));print('$v',d['value'],d['value_median'],d['value_max'],d['config']['stage_ms_per_step'])"
done
