#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s32; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab ";LOAMX_ODOM_GROUPS=3;LOAMX_ODOM_GROUPS=1" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
timeout 600 python bench.py --streams 4 --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab ";LOAMX_ODOM_GROUPS=1" > $out/ab4.json 2> $out/ab4.err
echo "--- 4 streams"; grep "^\[ab\]" $out/ab4.err
