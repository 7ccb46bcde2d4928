#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s36; mkdir -p $out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_s36
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s36 -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 1 > $out/bench.json 2> $out/err.txt
kt=$(find /tmp/prof_s36 -name '*kernel_trace.csv' | head -1)
python $root/scripts/prof_chain.py "$kt" v k_pose_init > $out/chain.txt 2>&1
cat $out/chain.txt | head -120
