#!/bin/bash
# Run on the MI355X box (through gpurun): rocprofv3 kernel trace of the exact driver bench command, per-kernel summary, per-chain
# timeline of one steady-state step.  Usage: scripts/gpu_profile.sh <tag> [extra bench args]
# Outputs under gpurun_out/<tag>/ : kernel_stats.csv  kernels.txt  chain.txt  bench.json
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 1 "$@" > $out/bench_profiled.json 2> $out/prof.err
kt=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
ks=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$ks" $out/kernel_stats.csv 2>/dev/null
python $root/scripts/prof_kernels.py "$kt" 40 > $out/kernels.txt 2>&1
python $root/scripts/prof_chain.py "$kt" v > $out/chain.txt 2>&1
python $root/scripts/prof_chain.py "$kt" v k_pose_init > $out/chain_by_registration.txt 2>&1   # (the window between two registrations' first kernels)
cd $root
if [ -z "${SKIP_PLAIN_BENCH:-}" ]; then
  timeout 300 python bench.py --steps 20 --warmup 5 "$@" > $out/bench.json 2> $out/bench.err
  tail -c 3000 $out/bench.json
fi
