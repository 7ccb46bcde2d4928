#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s57; mkdir -p $out
export TMPDIR=/tmp
for v in 1 2; do timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline > $out/live_$v.json 2> $out/live_$v.err; tail -2 $out/live_$v.err; python -c "
import json;d=json.load(open('$out/live_$v.json'));print(d['value'],d.get('value_nodes_concurrent'),d['ms_per_step'],d['config']['stage_ms_per_sweep'],d['roofline']['avg_launch_us'],d['roofline']['launches'])"; done
timeout 300 python bench.py --mode live --sensor HDL-32 --map-points 500000 --steps 60 --warmup 10 --no-cpu-baseline > $out/live_hdl32.json 2> $out/live_hdl32.err; python -c "
import json;d=json.load(open('$out/live_hdl32.json'));print(d['value'],d.get('value_nodes_concurrent'),d['ms_per_step'],d['config']['stage_ms_per_sweep'])"
