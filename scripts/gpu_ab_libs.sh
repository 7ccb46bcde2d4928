#!/bin/bash
# gpu_ab_libs.sh <tag> <rounds> <lib or '-' ...>: A/B of whole-library variants on ONE box — the bench's resident window
# (20 timed steps, 3 windows per process) for every library in turn, <rounds> times over (boxes differ by +-3 %, processes on one
# box by ~1 %: only lines of one call compare).  '-' = the product library.  Optional env: AB_TESTS="tests/a.py tests/b.py" runs
# those GPU tests with the LAST library first; AB_ARGS = extra bench arguments.
set -u
tag=$1; rounds=$2; shift 2
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
last=""
for l in "$@"; do last=$l; done
if [ -n "${AB_TESTS:-}" ]; then
  if [ "$last" != "-" ]; then export LOAMX_LIB=$root/$last; fi
  timeout 1500 python -m pytest $AB_TESTS -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc $?" >> $out/tests.log; tail -5 $out/tests.log
  unset LOAMX_LIB
fi
for r in $(seq 1 $rounds); do
  for l in "$@"; do
    name=$(basename $l .so)
    if [ "$l" = "-" ]; then unset LOAMX_LIB; name=product; else export LOAMX_LIB=$root/$l; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 3 --long-steps 0 ${AB_ARGS:-} > $out/bench_${name}_$r.json 2> $out/err_${name}_$r.txt
    python - <<PY
import json
try:
    d = json.load(open('$out/bench_${name}_$r.json'))
    print('%-28s r$r  value %8.0f  median %8.0f  max %8.0f  stage %s' % ('$name', d['value'], d.get('value_median', 0), d.get('value_max', 0), d['config'].get('stage_ms_per_step')))
except Exception as e:
    print('$name r$r FAILED', e)
PY
  done
done
