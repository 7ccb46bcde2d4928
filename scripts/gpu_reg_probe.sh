#!/bin/bash
# stand-alone kernel durations of the registration chain: scripts/gpu_reg_probe.sh <tag> [probe args]
set -u
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_$tag
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- python $root/scripts/gpu_reg_probe.py "$@" > $out/probe.txt 2> $out/probe.err
kt=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
python $root/scripts/trace_summary.py "$kt" 40 > $out/summary.txt 2>&1
cat $out/probe.txt; head -24 $out/summary.txt; tail -42 $out/summary.txt
