#!/bin/bash
# round 6: the linked odometry takes the less-flat cloud late (the per-ring voxel grid beside the first launch pair) — tests + A/B
set -u
root=$(pwd); out=$root/gpurun_out/r06_latelf; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_linked.py tests/test_gpu_nodes.py -x -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
tail -5 $out/tests.log
live() {  # name sensor map_points
  timeout 300 python bench.py --mode live --sensor $2 --map-points $3 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/$1.json 2> $out/$1.err
  python -c "
import json; d=json.load(open('$out/$1.json')); print('$1', d['value'], d['config']['stage_ms_per_sweep'], d['pose_err_vs_oracle']['mapped_pose']['max_m'] if 'pose_err_vs_oracle' in d else '')"
}
for r in 1 2 3; do
  unset LOAMX_LINK_NO_SPLIT; live split_vlp16_$r VLP-16 200000
  export LOAMX_LINK_NO_SPLIT=1; live nosplit_vlp16_$r VLP-16 200000
  unset LOAMX_LINK_NO_SPLIT; live split_hdl32_$r HDL-32 500000
  export LOAMX_LINK_NO_SPLIT=1; live nosplit_hdl32_$r HDL-32 500000
done
