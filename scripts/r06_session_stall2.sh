#!/bin/bash
# round 6: the cube histogram comes down by a kernel store (the hipMemcpyAsync blocked for milliseconds now and then) — VLP-16 processes + tests
set -u
root=$(pwd); out=$root/gpurun_out/r06_stall2; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_linked.py tests/test_gpu_mapping.py tests/test_gpu_nodes.py tests/test_gpu_batch.py -x -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
tail -3 $out/tests.log
for r in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 300 python bench.py --mode live --sensor VLP-16 --map-points 200000 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/vlp16_$r.json 2> $out/vlp16_$r.err
  python -c "
import json; d=json.load(open('$out/vlp16_$r.json')); print('vlp16_$r', d['value'], d['config']['stage_ms_per_sweep'])"
done
