#!/bin/bash
# A/B matrix inside ONE gpurun call (boxes of the pool differ by several percent, variants are only comparable within a call):
# scripts/gpu_matrix.sh <tag> "<env assignments of variant 1>" "<variant 2>" ...   ("" = defaults)
set -u
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
k=0
for v in "$@"; do
  k=$((k+1))
  env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie > $out/m$k.json 2> $out/m$k.err
  python - "$out/m$k.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d["config"]["stage_ms_per_step"]
    print(f"{sys.argv[2] or 'defaults':45s} {d['value']:9.1f} sweeps/s  {d['ms_per_step']:.4f} ms/step  F {s['features']:.3f} O {s['odometry']:.3f} M {s['registration']:.3f}  gn {d['roofline']['avg_launch_us']:.1f} us x {d['roofline']['launches']}")
except Exception as e:
    print(sys.argv[2], ": bench line unreadable:", e)
PY
done
