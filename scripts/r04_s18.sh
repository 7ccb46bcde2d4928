#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s18; mkdir -p $out
export TMPDIR=/tmp
scripts/gpu_trace_raw.sh r04_s18/eng > /dev/null 2>&1
head -16 $out/eng/summary.txt
tail -90 $out/eng/summary.txt
