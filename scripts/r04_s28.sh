#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s28; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_voxbucket.py -m gpu -x -q 2>&1 | tail -2
AB=";LOAMX_ODOM_GROUPS=1;LOAMX_ODOM_GROUPS=4;LOAMX_ODOM_ENGINE=1;LOAMX_ODOM_ENGINE=1 LOAMX_ODOM_ENGINES=1;LOAMX_BENCH_HANDLES=2;LOAMX_BENCH_HANDLES=2 LOAMX_ODOM_GROUPS=1"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab "$AB" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
