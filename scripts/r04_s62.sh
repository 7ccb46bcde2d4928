#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s62; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nodes.py tests/test_gpu_ingest.py tests/test_gpu_mapping.py tests/test_gpu_features.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; tail -3 $out/tests.log
for v in 1 2; do timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline > $out/live_$v.json 2> $out/live_$v.err; python -c "
import json;d=json.load(open('$out/live_$v.json'));print(d['value'],d.get('value_nodes_concurrent'),d['ms_per_step'],d['config']['stage_ms_per_sweep'])"; done
timeout 300 python bench.py --mode live --sensor HDL-32 --map-points 500000 --steps 60 --warmup 10 --no-cpu-baseline > $out/live_hdl32.json 2> $out/live_hdl32.err; python -c "
import json;d=json.load(open('$out/live_hdl32.json'));print(d['value'],d.get('value_nodes_concurrent'),d['ms_per_step'],d['config']['stage_ms_per_sweep'])"
