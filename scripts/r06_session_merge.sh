#!/bin/bash
# round 6: the epoch merge — synchronous on the stepping thread against asynchronous on a worker thread, one GPU, 200 timed steps (epochs of 5 steps)
set -u
root=$(pwd); out=$root/gpurun_out/r06_merge; mkdir -p $out
export TMPDIR=/tmp
for mode in none sync async; do
  case $mode in none) extra="";; sync) extra="--map-epoch-steps 5 --epoch-merge";; async) extra="--map-epoch-steps 5 --epoch-merge-async";; esac
  timeout 900 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 2 $extra > $out/bench_$mode.json 2> $out/bench_$mode.err
  python - <<PY
import json
try:
    d = json.loads(open('$out/bench_$mode.json').read().strip().splitlines()[-1])
    print('$mode', d['value'], d['value_median'], d['ms_per_step'], json.dumps(d['config'].get('map_epoch_merge'))[:600], 'epochs swapped', d['config']['map_epochs_swapped'])
except Exception as e:
    print('$mode FAILED', e); print(open('$out/bench_$mode.err').read()[-1500:])
PY
done
