#!/bin/bash
# round 6, last session: the driver's command on the final code (-> profiles/bench_r06.json) and the GPU suite
set -u
root=$(pwd); out=$root/gpurun_out/r06final; mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err ) 2> $out/bench.time; echo "bench rc $?" >> $out/bench.time
tail -4 $out/bench.time
python - <<PY
import json
d = json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print(json.dumps(d['summary']))
print(d['config']['bench_phase_seconds'], d['value_long'].get('host_seconds'))
PY
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $out/gpu_tests.log; tail -2 $out/gpu_tests.log
