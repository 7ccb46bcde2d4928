#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s15; mkdir -p $out
export TMPDIR=/tmp
LOAMX_LIB=$root/build/prof/libloamx_vb.so LOAMX_NO_LOOKAHEAD=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pcie --repeat 1 > $out/vb.json 2> $out/vb.err
grep -A6 "k_vb_reduce" $out/vb.err | tail -21
