#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s30; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -4 $out/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 5 --ab ";LOAMX_NO_PRESTAGE=1;;LOAMX_NO_PRESTAGE=1" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
