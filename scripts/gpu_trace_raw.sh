#!/bin/bash
# Run on the MI355X box: kernel trace of a short bench run, raw csv kept (gpurun_out/<tag>/kernel_trace.csv) + host-side traces
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$tag
LOAMX_REG_TRACE=1 LOAMX_PIPE_TRACE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- python $root/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-configs --no-pcie --repeat 1 "$@" > $out/bench.json 2> $out/host_trace.txt
kt=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
python $root/scripts/trace_summary.py "$kt" ${TAIL:-120} > $out/summary.txt 2>&1
