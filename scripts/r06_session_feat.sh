#!/bin/bash
# round 6: k_feat_ring with the ring staged through LDS — bit-exactness (feature tests), stamps, A/B against the previous kernel at 8 and 32 streams, live mode
set -u
root=$(pwd); out=$root/gpurun_out/r06_feat; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; tail -4 $out/tests.log
for S in 8 32; do
  LOAMX_LIB=$root/build/prof/libloamx_proffeat.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs --no-pcie --repeat 1 --streams $S 2>&1 >/dev/null | grep "feat_ring ts" | tail -12 > $out/feat_stamps_$S.txt
  echo "stamps $S:"; tail -3 $out/feat_stamps_$S.txt
done
AB_ARGS="--streams 8" bash scripts/gpu_ab_libs.sh r06_feat_ab8 3 - build/prof/libloamx_featold.so
AB_ARGS="--streams 32" bash scripts/gpu_ab_libs.sh r06_feat_ab32 3 - build/prof/libloamx_featold.so
for l in product featold product featold; do
  if [ $l = product ]; then unset LOAMX_LIB; else export LOAMX_LIB=$root/build/prof/libloamx_featold.so; fi
  timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/live_$l.json 2> $out/live_$l.err
  python -c "
import json; d=json.load(open('$out/live_$l.json')); print('live VLP-16 $l', d['value'], d['config']['stage_ms_per_sweep'])"
done
