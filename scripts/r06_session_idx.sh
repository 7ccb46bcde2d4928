#!/bin/bash
# round 6: the index set-up folded into k_bb_count (one launch less in the tail of every odometry pass): GPU tests, A/B against the previous build, live mode
set -u
root=$(pwd); out=$root/gpurun_out/r06_idx; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; tail -3 $out/tests.log
bash scripts/gpu_ab_libs.sh r06_idx_ab 3 - build/prof/libloamx_idxnew.so build/prof/libloamx_idxold.so
for l in product idxold product idxold; do
  if [ $l = product ]; then unset LOAMX_LIB; else export LOAMX_LIB=$root/build/prof/libloamx_idxold.so; fi
  timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/live_$l.json 2> $out/live_$l.err
  python -c "
import json; d=json.load(open('$out/live_$l.json')); print('live VLP-16 $l', d['value'], d['config']['stage_ms_per_sweep'])"
done
