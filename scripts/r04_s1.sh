#!/bin/bash
# round 4, GPU session 1: the GPU suite on the per-group odometry chains, then the A/B matrix odometry groups x pipeline handles x HW queues
set -u
root=$(pwd); out=$root/gpurun_out/r04_s1; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -3 $out/tests.log
AB="LOAMX_ODOM_GROUPS=1;LOAMX_ODOM_GROUPS=2;LOAMX_ODOM_GROUPS=4;LOAMX_ODOM_GROUPS=8;LOAMX_BENCH_HANDLES=2 LOAMX_ODOM_GROUPS=1;LOAMX_BENCH_HANDLES=2 LOAMX_ODOM_GROUPS=2;LOAMX_BENCH_HANDLES=2 LOAMX_ODOM_GROUPS=4;LOAMX_BENCH_HANDLES=4 LOAMX_ODOM_GROUPS=1;LOAMX_BENCH_HANDLES=4 LOAMX_ODOM_GROUPS=2"
timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab "$AB" > $out/ab_q8.json 2> $out/ab_q8.err
grep "^\[ab\]" $out/ab_q8.err
AB2="LOAMX_ODOM_GROUPS=1;LOAMX_ODOM_GROUPS=4;LOAMX_ODOM_GROUPS=8;LOAMX_BENCH_HANDLES=2 LOAMX_ODOM_GROUPS=2;LOAMX_BENCH_HANDLES=2 LOAMX_ODOM_GROUPS=4;LOAMX_BENCH_HANDLES=4 LOAMX_ODOM_GROUPS=2"
GPU_MAX_HW_QUEUES=16 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab "$AB2" > $out/ab_q16.json 2> $out/ab_q16.err
echo "--- 16 HW queues"; grep "^\[ab\]" $out/ab_q16.err
tail -c 600 $out/ab_q8.err
