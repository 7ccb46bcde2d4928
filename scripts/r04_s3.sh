#!/bin/bash
# round 4, GPU session 3 (diagnostic): what does a kernel boundary cost the other chains?  extra empty launches / fewer pairs
set -u
root=$(pwd); out=$root/gpurun_out/r04_s3; mkdir -p $out
export TMPDIR=/tmp
AB="LOAMX_ODOM_GROUPS=1;LOAMX_ODOM_GROUPS=1 LOAMX_ODOM_NOOPS=20;LOAMX_ODOM_GROUPS=1 LOAMX_ODOM_NOOPS=60;LOAMX_ODOM_GROUPS=1 LOAMX_ODOM_MAXIT=10;LOAMX_ODOM_GROUPS=1 LOAMX_ODOM_MAXIT=5;LOAMX_ODOM_GROUPS=2;LOAMX_ODOM_GROUPS=2 LOAMX_ODOM_MAXIT=10;LOAMX_ODOM_GROUPS=4 LOAMX_ODOM_MAXIT=10;LOAMX_ODOM_GROUPS=4 LOAMX_ODOM_MAXIT=5;LOAMX_ODOM_GROUPS=8 LOAMX_ODOM_MAXIT=5"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab "$AB" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
