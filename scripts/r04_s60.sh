#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s60; mkdir -p $out
export TMPDIR=/tmp
for q in 4 8 16; do
GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --repeat 2 > $out/bench_$q.json 2> $out/err_$q.txt
python -c "
import json;d=json.load(open('$out/bench_$q.json'));print('queues $q', d['value'],d['value_median'], d['pcie_inclusive']['value'], d['pcie_inclusive']['value_windows'], d['pcie_inclusive']['host_ms_per_step'])"
done
