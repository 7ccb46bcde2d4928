#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s29; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_gpu_mapping.py tests/test_gpu_dist.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -3 $out/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab ";LOAMX_NO_PRESTAGE=1;" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
