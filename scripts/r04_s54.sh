#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s54; mkdir -p $out
export TMPDIR=/tmp
LOAMX_PIPE_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --repeat 1 > $out/bench.json 2> $out/trace.txt
grep "^\[pipe" $out/trace.txt | tail -27 | cut -c1-100
python -c "
import json;d=json.load(open('$out/bench.json'));print(d['value'],d['ms_per_step']);print(json.dumps(d['pcie_inclusive'])[:600])"
