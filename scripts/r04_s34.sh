#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s34; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -4 $out/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 --ab ";LOAMX_SCAN_3PASS=1;" > $out/ab.json 2> $out/ab.err
grep "^\[ab\]" $out/ab.err
for v in "" "LOAMX_SCAN_3PASS=1"; do
  env $v timeout 600 python bench.py --mode live --steps 40 --warmup 5 --no-pcie --no-cpu-baseline > $out/live.json 2> $out/live.err
  python -c "
import json;o=json.loads(open('$out/live.json').read().strip().splitlines()[-1]);print('live [$v]',o['value'],o['ms_per_step'],o['config'].get('stage_ms_per_sweep'))"
done
