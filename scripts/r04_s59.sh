#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s59; mkdir -p $out
export TMPDIR=/tmp
free -g | head -2; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 --long-steps 400 > $out/bench.json 2> $out/bench.err
echo "elapsed s $SECONDS"
python -c "
import json;d=json.load(open('$out/bench.json'));print(d['value'],d['value_median'],d['ms_per_step'],d.get('value_long'),d.get('pose_err_vs_oracle',{}) and d['pose_err_vs_oracle']['mapped_pose']);print(d['cpu_baseline']['value'], d['pcie_inclusive']['value'])"
