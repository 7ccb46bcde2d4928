#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s45; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_qr6.py tests/test_gpu_pipeline.py tests/test_gpu_odometry.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -15 $out/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --ab ";LOAMX_PRESTAGE=1;LOAMX_NO_MIRROR_POLL=1;LOAMX_PRESTAGE=1 LOAMX_NO_MIRROR_POLL=1;" > $out/bench.json 2> $out/bench.err
grep "\[ab\]" $out/bench.err | tail -12
