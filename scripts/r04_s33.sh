#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s33; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python bench.py --mode live --steps 40 --warmup 5 --no-pcie > $out/live_vlp16.json 2> $out/live_vlp16.err
python -c "
import json;o=json.loads(open('$out/live_vlp16.json').read().strip().splitlines()[-1]);print(o['value'],o['ms_per_step'],o['config'].get('stage_ms_per_sweep'),o.get('cpu_baseline',{}).get('value'))"
cd /tmp; rm -rf /tmp/prof_live
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_live -- python $root/bench.py --mode live --steps 30 --warmup 5 --no-pcie --no-cpu-baseline > $out/live_prof.json 2> $out/live_prof.err
ks=$(find /tmp/prof_live -name '*kernel_stats.csv' | head -1); cp "$ks" $out/live_kernel_stats.csv
head -40 $out/live_kernel_stats.csv | cut -c1-150
