#!/bin/bash
# round 6: the mapping's parameter block in the arguments of the single-sweep gather (k_gather_one) — tests + live figures against HEAD's library
set -u
root=$(pwd); out=$root/gpurun_out/r06_gone; mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_linked.py tests/test_gpu_mapping.py tests/test_gpu_nodes.py -x -q > $out/tests.log 2>&1; echo "tests rc $?" | tee -a $out/tests.log
tail -3 $out/tests.log
live() {  # name sensor map_points
  timeout 300 python bench.py --mode live --sensor $2 --map-points $3 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/$1.json 2> $out/$1.err
  python -c "
import json; d=json.load(open('$out/$1.json')); print('$1', d['value'], d['config']['stage_ms_per_sweep'])"
}
for r in 1 2 3 4; do
  unset LOAMX_LIB; live new_vlp16_$r VLP-16 200000
  export LOAMX_LIB=$root/build/head/loam/libloamx.so; live old_vlp16_$r VLP-16 200000
  unset LOAMX_LIB; live new_hdl32_$r HDL-32 500000
  export LOAMX_LIB=$root/build/head/loam/libloamx.so; live old_hdl32_$r HDL-32 500000
done
