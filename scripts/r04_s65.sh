#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s65; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_odometry.py tests/test_gpu_pipeline.py tests/test_gpu_nodes.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; tail -4 $out/tests.log
for v in 1 0 1; do
LOAMX_BB_SETUP_FUSED=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 > $out/bench_$v.json 2> $out/err_$v.txt
python -c "
import json;d=json.load(open('$out/bench_$v.json'));print('fused $v', d['value'],d['value_median'],d['value_max'],d['config']['stage_ms_per_step'])"
done
