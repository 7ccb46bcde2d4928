#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s43; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -5 $out/tests.log
LOAMX_BENCH_LOOK=6 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --ab "LOAMX_ODOM_AHEAD=2;LOAMX_ODOM_AHEAD=3;LOAMX_ODOM_AHEAD=4;LOAMX_ODOM_AHEAD=6;LOAMX_ODOM_AHEAD=4 LOAMX_PRESTAGE=1;LOAMX_ODOM_AHEAD=6 LOAMX_PRESTAGE=1;LOAMX_ODOM_AHEAD=2" > $out/bench.json 2> $out/bench.err
grep "\[ab\]" $out/bench.err | tail -12
