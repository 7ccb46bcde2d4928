#!/bin/bash
set -u
root=$(pwd); out=$root/gpurun_out/r04_s64; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_features.py tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_gpu_nodes.py -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; tail -6 $out/tests.log
LOAMX_NO_LOOKAHEAD=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --repeat 1 > $out/seq.json 2> $out/seq.err
python -c "
import json;d=json.load(open('$out/seq.json'));print('sequential stage ms',d['config']['stage_ms_per_step'])"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 3 > $out/bench.json 2> $out/bench.err
python -c "
import json;d=json.load(open('$out/bench.json'));print(d['value'],d['value_median'],d['value_max'],d['config']['stage_ms_per_step'])"
