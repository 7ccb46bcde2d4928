#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s7; mkdir -p $out
export TMPDIR=/tmp
LOAMX_NO_LOOKAHEAD=1 scripts/gpu_trace_raw.sh r04_s7/seq_lds > /dev/null 2>&1
LOAMX_NO_LOOKAHEAD=1 LOAMX_ODOM_CORR_LEGACY=1 scripts/gpu_trace_raw.sh r04_s7/seq_legacy > /dev/null 2>&1
grep -E "k_odom" $out/seq_lds/summary.txt | head -3
grep -E "k_odom" $out/seq_legacy/summary.txt | head -3
