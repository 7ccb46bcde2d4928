#!/bin/bash
# round 6, session a: GPU tests with the adaptive pair enqueue, pair-mode A/B (batched + live), k_odom_lm without spills
set -u
root=$(pwd); out=$root/gpurun_out/r06a; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; tail -4 $out/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 5 --long-steps 0 \
  --ab "LOAMX_ODOM_PAIRS=all;;LOAMX_ODOM_PAIRS=exact;LOAMX_ODOM_PAIRS=all;;LOAMX_ODOM_PAIRS=exact" > $out/ab_pairs.json 2> $out/ab_pairs.txt
grep "^\[ab\]" $out/ab_pairs.txt
for m in all lag exact; do
  LOAMX_ODOM_PAIRS=$m timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline --no-side-configs --no-live-nodes > $out/live_vlp16_$m.json 2> $out/live_vlp16_$m.err
  python -c "
import json; d=json.load(open('$out/live_vlp16_$m.json')); print('live VLP-16 pairs=$m', d['value'], d['config']['stage_ms_per_sweep'])"
done
bash scripts/gpu_ab_libs.sh r06a_libs 2 - build/prof/libloamx_lmw3.so
