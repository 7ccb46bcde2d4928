#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s37; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -8 $out/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --ab "LOAMX_NO_MIRROR_POLL=1;;LOAMX_NO_MIRROR_POLL=1;" > $out/bench.json 2> $out/bench.err
grep -i "ab\b\|variant\|sweeps" $out/bench.err | tail -12; cat $out/bench.json
