#!/bin/bash
# gpu_trace_session.sh <tag> <lib or '-'> [ENV=VALUE ...]: the bench's resident window with the host-side step trace (LOAMX_PIPE_TRACE=1,
# no HIP-event sampling) -> gpurun_out/<tag>/trace.txt + summary (scripts/pipe_trace_summary.py): who bounds the step
set -u
tag=$1; lib=$2; shift 2
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ "$lib" != "-" ]; then export LOAMX_LIB=$root/$lib; fi
for kv in "$@"; do export "$kv"; done
LOAMX_PIPE_TRACE=1 LOAMX_BENCH_TIMING_PERIOD=1000 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 1 --long-steps 0 > $out/bench.json 2> $out/trace.txt
python scripts/pipe_trace_summary.py $out/trace.txt $out/bench.json | tee $out/summary.txt
