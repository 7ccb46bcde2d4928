#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s51; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_odometry.py tests/test_gpu_pipeline.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -5 $out/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --ab ";LOAMX_PRESTAGE=1;" > $out/bench.json 2> $out/bench.err
grep "\[ab\]" $out/bench.err | tail -12
python -c "
import json;d=json.load(open('$out/bench.json'));print(d['value'],d['value_median'],d['ms_per_step'],d['config']['stage_ms_per_step'], d.get('pose_err_vs_oracle'))"
