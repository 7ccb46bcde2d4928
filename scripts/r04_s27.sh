#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s27; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_odometry.py tests/test_gpu_next.py tests/test_gpu_pipeline.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -3 $out/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab ";" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
LOAMX_NO_LOOKAHEAD=1 scripts/gpu_trace_raw.sh r04_s27/seq > /dev/null 2>&1
grep -E "k_gn_iter|k_odom_lm|k_odom_corr" $out/seq/summary.txt | head -4
