#!/bin/bash
# Run on the MI355X box: three SEPARATE rocprofv3 --pmc passes of the bench command (FETCH_SIZE | WRITE_SIZE | TCC hit/miss), each with
# --kernel-trace only (never combined with other trace domains), then scripts/pmc_summary.py.
# usage: scripts/gpu_pmc.sh <tag> [bench args]     (default: the driver's workload, 4 steps; e.g. "--mode live --steps 8 --warmup 2 --no-cpu-baseline --no-side-configs --no-live-nodes")
set -u
tag=$1; shift
if [ $# -eq 0 ]; then set -- --steps 4 --warmup 1 --no-cpu-baseline --no-side-configs --no-pcie --repeat 1; fi
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for set in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum"; do
  name=${set%%:*}; ctr=${set#*:}
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$name -- python $root/bench.py "$@" > $out/pmc_$name.log 2>&1
done
python $root/scripts/pmc_summary.py $out/pmc_summary.json /tmp/pmc_fetch /tmp/pmc_write /tmp/pmc_l2 > $out/pmc_summary.log 2>&1
tail -5 $out/pmc_summary.log
