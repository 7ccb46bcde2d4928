#!/bin/bash
# round 6, session b: the driver's command with the new blocks (timing of every phase), the long-chain GPU test
set -u
root=$(pwd); out=$root/gpurun_out/r06b; mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err ) 2> $out/bench.time; echo "bench rc $?" >> $out/bench.time
tail -3 $out/bench.time; tail -5 $out/bench.err
python - <<PY
import json
d = json.load(open('$out/bench.json'))
print(json.dumps(d['summary'], indent=1))
print(d['config']['bench_phase_seconds'])
PY
timeout 900 python -m pytest tests/test_gpu_longchain.py -m gpu -x -q -s > $out/longchain.log 2>&1; tail -6 $out/longchain.log
