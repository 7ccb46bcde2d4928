#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s16; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_voxbucket.py tests/test_gpu_batch.py tests/test_gpu_pipeline.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -3 $out/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab ";" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
LOAMX_NO_LOOKAHEAD=1 scripts/gpu_trace_raw.sh r04_s16/seq > /dev/null 2>&1
grep -E "k_vb_reduce|k_gn_iter|k_vb_plan|k_vb_stack" $out/seq/summary.txt | head -6
