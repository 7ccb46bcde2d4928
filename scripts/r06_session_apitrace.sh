#!/bin/bash
# round 6: which runtime calls block their caller for > 300 us in steady state?  (-DLOAMX_API_TRACE build: build/apitrace)
set -u
root=$(pwd); out=$root/gpurun_out/r06_apitrace; mkdir -p $out
export TMPDIR=/tmp
export LOAMX_LIB=$root/build/apitrace/loam/libloamx.so
for r in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 40 --long-steps 0 > $out/batched_$r.json 2> $out/batched_$r.err
  python -c "
import json; d=json.load(open('$out/batched_$r.json')); print('batched r$r value %8.0f median %8.0f min %8.0f max %8.0f' % (d['value'], d['value_median'], d['value_min'], d['value_max']))"
  echo "api trace lines: $(grep -c 'api trace' $out/batched_$r.err)"
  grep -h "api trace" $out/batched_$r.err | sed 's/\[api trace\] [0-9]* us in //' | sort | uniq -c | sort -rn | head -12
  grep -h "api trace" $out/batched_$r.err | sort -t' ' -k3 -n -r | head -5
done
for cfg in "VLP-16 200000 vlp16" "HDL-32 500000 hdl32"; do
  set -- $cfg
  timeout 300 python bench.py --mode live --sensor $1 --map-points $2 --steps 200 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/live_$3.json 2> $out/live_$3.err
  python -c "
import json; d=json.load(open('$out/live_$3.json')); print('live $3', d['value'], d['config']['stage_ms_per_sweep'])"
  echo "api trace lines: $(grep -c 'api trace' $out/live_$3.err)"
  grep -h "api trace" $out/live_$3.err | sed 's/\[api trace\] [0-9]* us in //' | sort | uniq -c | sort -rn | head -12
done
