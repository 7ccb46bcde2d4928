#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s8; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_odometry.py tests/test_gpu_next.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -3 $out/tests.log
AB="LOAMX_ODOM_CORR_LEGACY=1;;LOAMX_ODOM_GROUPS=1 LOAMX_ODOM_CORR_LEGACY=1;LOAMX_ODOM_GROUPS=1"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab "$AB" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
LOAMX_NO_LOOKAHEAD=1 scripts/gpu_trace_raw.sh r04_s8/seq_lds > /dev/null 2>&1
grep -E "k_odom" $out/seq_lds/summary.txt | head -24
