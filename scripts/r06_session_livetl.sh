#!/bin/bash
# round 6: kernel timeline of the sequential-SLAM chain (last sweeps of a profiled run, queue ids) — where is the critical path now?
set -u
root=$(pwd); out=$root/gpurun_out/r06_livetl; mkdir -p $out
export TMPDIR=/tmp
for cfg in "VLP-16 200000 vlp16" "HDL-32 500000 hdl32"; do
  set -- $cfg
  cd /tmp; rm -rf /tmp/prof_tl
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python $root/bench.py --mode live --sensor $1 --map-points $2 --steps 20 --warmup 3 --no-cpu-baseline --no-side-configs --no-live-nodes > $out/bench_$3.json 2> $out/prof_$3.err
  kt=$(find /tmp/prof_tl -name '*kernel_trace.csv' | head -1)
  python $root/scripts/trace_summary.py "$kt" 400 > $out/timeline_$3.txt 2>&1
  cd $root
done
