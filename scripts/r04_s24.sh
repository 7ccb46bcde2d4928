#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s24; mkdir -p $out
export TMPDIR=/tmp
LOAMX_ODOM_ENGINES=1 LOAMX_OE_TRACE=1 LOAMX_PIPE_TRACE=1 timeout 600 python bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-pcie --repeat 1 > $out/bench.json 2> $out/host_trace.txt
grep -E "\[oe|pipe t=" $out/host_trace.txt | tail -90 | cut -c1-150
