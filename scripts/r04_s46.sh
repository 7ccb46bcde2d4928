#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s46; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -15 $out/tests.log
