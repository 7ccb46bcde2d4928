#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s53; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log
tail -6 $out/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python -c "
import json;d=json.load(open('$out/bench.json'));print(d['value'],d['value_median'],d['ms_per_step']);print(json.dumps(d['pcie_inclusive'])[:900])"
