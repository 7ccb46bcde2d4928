#!/bin/bash
# One gpurun call of this round's routine: GPU test suite, the driver's bench command, A/B variants through environment
# switches, kernel trace + per-chain timeline.  Usage: scripts/gpu_session.sh <tag> [variants...]; outputs under gpurun_out/<tag>/
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -40 ) > $out/tests.log
tail -5 $out/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - "$out/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bench:", d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"], "pcie", d.get("pcie_inclusive", {}).get("value"), "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
except Exception as e:
    print("bench line unreadable:", e)
PY
for v in "$@"; do
  name=$(echo "$v" | tr '= ' '__')
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie > $out/bench_$name.json 2> $out/bench_$name.err
  python - "$out/bench_$name.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], ":", d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"])
except Exception as e:
    print(sys.argv[2], ": bench line unreadable:", e)
PY
done
scripts/gpu_profile.sh ${tag}_prof > /dev/null 2>&1
head -60 $root/gpurun_out/${tag}_prof/chain.txt
head -30 $root/gpurun_out/${tag}_prof/kernels.txt
