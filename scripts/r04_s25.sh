#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s25; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -s -k "bench_scale or toggled" > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -12 $out/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-pcie --repeat 2 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python - <<'PY'
import json
o=json.loads(open("gpurun_out/r04_s25/bench.json").read().strip().splitlines()[-1])
print(o["value"], o["ms_per_step"], json.dumps(o.get("pose_err_vs_oracle"))[:900])
PY
