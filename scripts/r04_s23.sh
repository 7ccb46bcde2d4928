#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s23; mkdir -p $out
export TMPDIR=/tmp
LOAMX_ODOM_ENGINES=1 LOAMX_PIPE_TRACE=1 LOAMX_REG_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 1 > $out/bench.json 2> $out/host_trace.txt
grep "pipe t=" $out/host_trace.txt | tail -12 | cut -c1-140
grep "\[reg\]" $out/host_trace.txt | tail -4
