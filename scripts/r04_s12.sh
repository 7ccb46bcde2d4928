#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s12; mkdir -p $out
export TMPDIR=/tmp
PROBE_SHORT=1 LOAMX_LIB=$root/build/prof/libloamx_gn.so timeout 600 python scripts/gpu_gn_probe.py 8 > $out/probe.txt 2> $out/probe.err
grep -E "solve_sweep|gn iter 0" $out/probe.err | tail -4 | cut -c1-1500
tail -5 $out/probe.txt
AB="LOAMX_ODOM_CELL=1.05;LOAMX_ODOM_CELL=1.4;;LOAMX_ODOM_CELL=2.8"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab "$AB" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
