#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s44; mkdir -p $out
export TMPDIR=/tmp
LOAMX_BENCH_LOOK=12 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --ab "LOAMX_ODOM_AHEAD=6;LOAMX_ODOM_AHEAD=8;LOAMX_ODOM_AHEAD=12;LOAMX_ODOM_AHEAD=4;LOAMX_ODOM_AHEAD=8 LOAMX_PRESTAGE=1;LOAMX_ODOM_AHEAD=6" > $out/bench.json 2> $out/bench.err
grep "\[ab\]" $out/bench.err | tail -12
LOAMX_BENCH_LOOK=12 timeout 900 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-pcie --ab "LOAMX_ODOM_AHEAD=2;LOAMX_ODOM_AHEAD=4;LOAMX_ODOM_AHEAD=6;LOAMX_ODOM_AHEAD=8;LOAMX_ODOM_AHEAD=12" > $out/bench60.json 2> $out/bench60.err
grep "\[ab\]" $out/bench60.err | tail -12
