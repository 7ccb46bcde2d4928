#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s39; mkdir -p $out
export TMPDIR=/tmp
LOAMX_PIPE_TRACE=1 LOAMX_BENCH_TIMING_PERIOD=1000 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 1 > $out/bench.json 2> $out/trace.txt
grep "^\[pipe" $out/trace.txt | tail -22
LOAMX_PRESTAGE=1 LOAMX_PIPE_TRACE=1 LOAMX_BENCH_TIMING_PERIOD=1000 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 1 > $out/bench2.json 2> $out/trace2.txt
grep "^\[pipe" $out/trace2.txt | tail -12
