#!/bin/bash
# round 4, GPU session 2 (diagnostic): why do 4 odometry chains / 2 handles run slower than 1-2?  kernel traces + host traces
set -u
root=$(pwd); out=$root/gpurun_out/r04_s2; mkdir -p $out
export TMPDIR=/tmp
for g in 1 2 4; do
  LOAMX_ODOM_GROUPS=$g scripts/gpu_trace_raw.sh r04_s2/g$g > /dev/null 2>&1
  kt=$(find /tmp/prof_r04_s2/g$g -name '*kernel_trace.csv' | head -1)
  python scripts/prof_chain.py "$kt" v > $out/g$g/chain.txt 2>&1
  grep -c . $out/g$g/host_trace.txt
done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print(len(os.sched_getaffinity(0)))"
AB="LOAMX_ODOM_GROUPS=4;LOAMX_ODOM_GROUPS=4 LOAMX_SPIN_US=0;LOAMX_ODOM_GROUPS=4 LOAMX_PRIO_ODOM=0 LOAMX_PRIO_FEAT=0;LOAMX_ODOM_GROUPS=2"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab "$AB" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
