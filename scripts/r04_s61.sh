#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s61; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "streaming or raw" > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; tail -4 $out/tests.log
for v in 1 0; do
LOAMX_H2D_DIRECT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --repeat 2 > $out/bench_$v.json 2> $out/err_$v.txt
python -c "
import json;d=json.load(open('$out/bench_$v.json'));print('direct $v', d['value'],d['value_median'], d['pcie_inclusive'].get('value'), d['pcie_inclusive'].get('value_windows'), d['pcie_inclusive'].get('host_ms_per_step'), d['pcie_inclusive'].get('error'))"
done
