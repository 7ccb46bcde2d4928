#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s52; mkdir -p $out
export TMPDIR=/tmp
LOAMX_PIPE_TRACE=1 LOAMX_BENCH_TIMING_PERIOD=1000 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pcie --repeat 1 > $out/bench.json 2> $out/trace.txt
grep "^\[pipe" $out/trace.txt | tail -42 | cut -c1-110
