#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s38; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --ab ";LOAMX_PRESTAGE=1;LOAMX_PRESTAGE=1 LOAMX_NO_MIRROR_POLL=1;LOAMX_BENCH_TIMING_PERIOD=1000;LOAMX_BENCH_TIMING_PERIOD=1000 LOAMX_PRESTAGE=1;LOAMX_BENCH_TIMING_PERIOD=1000 LOAMX_NO_MIRROR_POLL=1;;LOAMX_PRESTAGE=1" > $out/bench.json 2> $out/bench.err
grep "\[ab\]" $out/bench.err | tail -12
