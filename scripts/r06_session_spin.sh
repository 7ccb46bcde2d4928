#!/bin/bash
# round 6: polling waits (spin_sync / spin_event) instead of the runtime's blocking waits: GPU suite, batched A/B against the previous build, live runs
set -u
root=$(pwd); out=$root/gpurun_out/r06_spin; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_linked.py tests/test_gpu_mapping.py tests/test_gpu_nodes.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; tail -3 $out/tests.log

for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline > $out/live_$i.json 2> $out/live_$i.err
  python -c "
import json; d=json.load(open('$out/live_$i.json')); print('live VLP-16 run $i', d['value'], d.get('value_nodes_concurrent'), d['config']['host_message_chain']['sweeps_per_s'], d['config']['stage_ms_per_sweep'])"
done
for i in 1 2 3; do
  timeout 300 python bench.py --mode live --sensor HDL-32 --map-points 500000 --steps 100 --warmup 10 --no-cpu-baseline > $out/live32_$i.json 2> $out/live32_$i.err
  python -c "
import json; d=json.load(open('$out/live32_$i.json')); print('live HDL-32 run $i', d['value'], d.get('value_nodes_concurrent'), d['config']['stage_ms_per_sweep'])"
done
