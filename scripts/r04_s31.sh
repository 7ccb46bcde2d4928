#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s31; mkdir -p $out
export TMPDIR=/tmp
SKIP_PLAIN_BENCH=1 scripts/gpu_profile.sh r04_s31/prof > /dev/null 2>&1
cat $out/prof/chain.txt | head -90
