#!/bin/bash
# sequential-SLAM mode (BASELINE configs[1] / [2]): bench line + rocprofv3 kernel trace summary.  usage: scripts/gpu_live_profile.sh <tag> [bench args]
set -u
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $root/bench.py --mode live --steps 20 --warmup 3 --no-cpu-baseline --no-side-configs "$@" > $out/bench_profiled.json 2> $out/prof.err
ks=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
kt=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
cp "$ks" $out/kernel_stats.csv 2>/dev/null
python $root/scripts/trace_summary.py "$kt" 150 > $out/summary.txt 2>&1
cd $root
timeout 600 python bench.py --mode live --steps 100 --warmup 10 "$@" > $out/bench.json 2> $out/bench.err
tail -c 1500 $out/bench.json; echo; head -28 $out/summary.txt
