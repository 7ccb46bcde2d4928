#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s26; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 --no-pcie --repeat 2 > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err
python - <<'PY'
import json
o=json.loads(open("gpurun_out/r04_s26/bench.json").read().strip().splitlines()[-1])
print(o["value"], o["ms_per_step"], json.dumps(o.get("pose_err_vs_oracle"))[:700])
PY
