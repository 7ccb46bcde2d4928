#!/bin/bash
# round 6: the saturated device (32 streams per GPU): kernel stats + per-queue timeline, PMC passes, host-side step trace, k_feat_ring stamps
set -u
root=$(pwd); out=$root/gpurun_out/r06_sat32; mkdir -p $out
export TMPDIR=/tmp
SKIP_PLAIN_BENCH=1 bash scripts/gpu_profile.sh r06_sat32_prof --streams 32 > /dev/null 2>&1
cp gpurun_out/r06_sat32_prof/kernel_stats.csv $out/kernel_stats.csv; cp gpurun_out/r06_sat32_prof/kernels.txt $out/kernels_per_step.txt; cp gpurun_out/r06_sat32_prof/chain.txt $out/chain_timeline.txt
cp gpurun_out/r06_sat32_prof/bench_profiled.json $out/bench_profiled.json
head -25 $out/kernels_per_step.txt
bash scripts/gpu_pmc.sh r06_sat32_pmc --steps 4 --warmup 1 --no-cpu-baseline --no-side-configs --no-pcie --repeat 1 --streams 32 > $out/pmc.log 2>&1
cp gpurun_out/r06_sat32_pmc/pmc_summary.json $out/pmc_summary.json 2>/dev/null; tail -3 $out/pmc.log
# unprofiled: who bounds the step at 32 streams
LOAMX_PIPE_TRACE=1 LOAMX_BENCH_TIMING_PERIOD=1000 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 1 --long-steps 0 --streams 32 > $out/trace_bench.json 2> $out/trace.txt
python scripts/pipe_trace_summary.py $out/trace.txt $out/trace_bench.json | tee $out/pipe_trace_summary.txt
# plain windows, 8 / 16 / 32 / 48 / 64 streams (same box)
for S in 8 16 32 48 64; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 3 --streams $S > $out/streams_$S.json 2> $out/streams_$S.err
  python -c "import json,sys; d=json.loads(open('$out/streams_$S.json').read().strip().splitlines()[-1]); print('streams', $S, d['value'], d['value_median'], d['ms_per_step'], d['config']['stage_ms_per_step'], 'path_hbm_frac', d['config']['path_hbm_frac'])"
done
# k_feat_ring in-kernel stamps (-DLOAMX_PROF_FEAT build), 8 and 32 streams
if [ -f build/prof/libloamx_proffeat.so ]; then
  for S in 8 32; do
    LOAMX_LIB=$root/build/prof/libloamx_proffeat.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs --no-pcie --repeat 1 --streams $S 2>&1 >/dev/null | grep "feat_ring ts" | tail -12 > $out/feat_stamps_$S.txt
    echo "stamps $S:"; tail -4 $out/feat_stamps_$S.txt
  done
fi
