#!/bin/bash
# round 6: the surround cloud cut behind the prepared partition (the next sweep waits for the partition only) — tests + A/B against the
# library with the old order (build/prof/libloamx_surfirst.so), both with the late less-flat hand-over
set -u
root=$(pwd); out=$root/gpurun_out/r06_surround; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_linked.py tests/test_gpu_mapping.py tests/test_gpu_nodes.py -x -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
tail -5 $out/tests.log
live() {  # name sensor map_points
  timeout 300 python bench.py --mode live --sensor $2 --map-points $3 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/$1.json 2> $out/$1.err
  python -c "
import json; d=json.load(open('$out/$1.json')); print('$1', d['value'], d['config']['stage_ms_per_sweep'])"
}
for r in 1 2 3 4; do
  unset LOAMX_LIB; live new_vlp16_$r VLP-16 200000
  export LOAMX_LIB=$root/build/prof/libloamx_surfirst.so; live old_vlp16_$r VLP-16 200000
  unset LOAMX_LIB; live new_hdl32_$r HDL-32 500000
  export LOAMX_LIB=$root/build/prof/libloamx_surfirst.so; live old_hdl32_$r HDL-32 500000
done
unset LOAMX_LIB
LOAMX_MAP_TRACE=1 timeout 300 python bench.py --mode live --sensor VLP-16 --map-points 200000 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/trace_vlp16.json 2> $out/trace_vlp16.err
grep "mean of" $out/trace_vlp16.err | tail -3
LOAMX_MAP_TRACE=1 timeout 300 python bench.py --mode live --sensor HDL-32 --map-points 500000 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/trace_hdl32.json 2> $out/trace_hdl32.err
grep "mean of" $out/trace_hdl32.err | tail -3
