#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s22; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -3 $out/tests.log
AB=";LOAMX_ODOM_ENGINES=1;LOAMX_ODOM_ENGINES=4;LOAMX_ODOM_ENGINE=0"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab "$AB" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
