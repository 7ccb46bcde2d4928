#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s55; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mapping.py tests/test_gpu_pipeline.py -m gpu -x -q -s -k "epoch or streaming or lookahead_is" > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -25 $out/tests.log
