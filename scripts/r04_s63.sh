#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s63; mkdir -p $out
export TMPDIR=/tmp
LOAMX_LIB=build/prof/libloamx_feat.so LOAMX_NO_LOOKAHEAD=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --repeat 1 > $out/bench.json 2> $out/err.txt
grep "feat_ring ts" $out/err.txt | tail -8
