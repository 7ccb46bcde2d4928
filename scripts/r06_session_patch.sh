#!/bin/bash
# round 6: the late offsets travel with the next correspondence launch instead of a copy command — tests + live figures
set -u
root=$(pwd); out=$root/gpurun_out/r06_patch; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_linked.py tests/test_gpu_odometry.py tests/test_gpu_nodes.py tests/test_gpu_pipeline.py -x -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
tail -5 $out/tests.log
live() {  # name sensor map_points
  timeout 300 python bench.py --mode live --sensor $2 --map-points $3 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/$1.json 2> $out/$1.err
  python -c "
import json; d=json.load(open('$out/$1.json')); print('$1', d['value'], d['config']['stage_ms_per_sweep'])"
}
for r in 1 2 3 4; do
  live new_vlp16_$r VLP-16 200000
  live new_hdl32_$r HDL-32 500000
done
