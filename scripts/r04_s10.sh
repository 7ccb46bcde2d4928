#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s10; mkdir -p $out
export TMPDIR=/tmp
LOAMX_LIB=$root/build/prof/libloamx_corr.so LOAMX_NO_LOOKAHEAD=1 LOAMX_ODOM_GROUPS=1 timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-pcie --repeat 1 > $out/bench.json 2> $out/err.txt
grep "corr ts" $out/err.txt | tail -18
