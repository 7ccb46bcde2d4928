#!/bin/bash
# round 6: speculation on one launch (lag) against one pair (lag2) against none (exact) against all; GPU tests first
set -u
root=$(pwd); out=$root/gpurun_out/r06_pairs2; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_odometry.py tests/test_gpu_linked.py tests/test_gpu_batch.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; tail -3 $out/tests.log
timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 5 --long-steps 0 \
  --ab "LOAMX_ODOM_PAIRS=lag2;;LOAMX_ODOM_PAIRS=all;LOAMX_ODOM_PAIRS=lag2;;LOAMX_ODOM_PAIRS=exact;LOAMX_ODOM_PAIRS=lag2;" > $out/ab.json 2> $out/ab.txt
grep "^\[ab\]" $out/ab.txt
for m in lag2 lag lag2 lag; do
  LOAMX_ODOM_PAIRS=$m timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/live_$m.json 2> $out/live_$m.err
  python -c "
import json; d=json.load(open('$out/live_$m.json')); print('live VLP-16 pairs=$m', d['value'], d['config']['stage_ms_per_sweep'])"
done
