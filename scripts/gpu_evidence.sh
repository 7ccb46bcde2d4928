#!/bin/bash
# The round's evidence in one gpurun call: GPU test log, the driver's bench command, kernel trace + chain timeline, PMC passes,
# sequential (stand-alone) kernel durations, the sequential-SLAM configurations, the stream sweep, a copy trace of the PCIe window.
# usage: scripts/gpu_evidence.sh <tag>     outputs: gpurun_out/<tag>/...
set -u
tag=$1
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -30 ) > $out/gpu_tests.log; tail -2 $out/gpu_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $out/smoke.log; tail -1 $out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 400 $out/bench.json; echo
SKIP_PLAIN_BENCH=1 scripts/gpu_profile.sh ${tag}_prof > /dev/null 2>&1; cp gpurun_out/${tag}_prof/kernel_stats.csv gpurun_out/${tag}_prof/kernels.txt gpurun_out/${tag}_prof/chain.txt gpurun_out/${tag}_prof/chain_by_registration.txt $out/ 2>/dev/null
scripts/gpu_pmc.sh ${tag}_pmc > $out/pmc.log 2>&1; cp gpurun_out/${tag}_pmc/pmc_summary.json $out/ 2>/dev/null; tail -3 $out/pmc.log
LOAMX_NO_LOOKAHEAD=1 scripts/gpu_trace_raw.sh ${tag}_seq > /dev/null 2>&1; cp gpurun_out/${tag}_seq/summary.txt $out/sequential_summary.txt 2>/dev/null
scripts/gpu_live_profile.sh ${tag}_live1 > $out/live_vlp16.log 2>&1; tail -c 300 gpurun_out/${tag}_live1/bench.json; echo
scripts/gpu_live_profile.sh ${tag}_live2 --sensor HDL-32 --map-points 500000 > $out/live_hdl32.log 2>&1; tail -c 300 gpurun_out/${tag}_live2/bench.json; echo
scripts/gpu_pmc.sh ${tag}_pmc_live1 --mode live --steps 8 --warmup 2 --no-cpu-baseline --no-side-configs --no-live-nodes > $out/pmc_live_vlp16.log 2>&1; cp gpurun_out/${tag}_pmc_live1/pmc_summary.json $out/live_vlp16_pmc_summary.json 2>/dev/null
scripts/gpu_pmc.sh ${tag}_pmc_live2 --mode live --sensor HDL-32 --map-points 500000 --steps 8 --warmup 2 --no-cpu-baseline --no-side-configs --no-live-nodes > $out/pmc_live_hdl32.log 2>&1; cp gpurun_out/${tag}_pmc_live2/pmc_summary.json $out/live_hdl32_pmc_summary.json 2>/dev/null
scripts/gpu_trace_session.sh ${tag}_trace - > $out/pipe_trace.log 2>&1; cp gpurun_out/${tag}_trace/summary.txt $out/pipe_trace_summary.txt 2>/dev/null
for S in 1 4 16 32; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs --no-pcie --repeat 2 --streams $S > $out/streams_$S.json 2> $out/streams_$S.err
  python -c "import json,sys; d=json.loads(open('$out/streams_$S.json').read().strip().splitlines()[-1]); print('streams', $S, d['value'], d['value_median'], d['ms_per_step'])"
done
cd /tmp; rm -rf /tmp/prof_copy
LOAMX_PIPE_TRACE=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_copy -- python $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --repeat 1 > $out/pcie_traced.json 2> $out/pcie_traced.err
mc=$(find /tmp/prof_copy -name '*memory_copy_trace.csv' | head -1)
python - "$mc" > $out/pcie_copies.txt 2>&1 <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("memory copies:", len(rows))
big = [r for r in rows if int(r.get("Bytes", r.get("Size", 0)) or 0) > 1 << 20]
agg = collections.defaultdict(list)
for r in big:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[r.get("Direction", "?")].append((int(r.get("Bytes", r.get("Size", 0))), dur))
for k, v in agg.items():
    b = sum(x for x, _ in v); t = sum(y for _, y in v)
    print(k, "copies >1MiB:", len(v), "mean MiB", round(b / len(v) / 2**20, 2), "mean us", round(t / len(v), 1), "GB/s", round(b / t / 1e3, 1))
PY
cat $out/pcie_copies.txt
kt=$(find /tmp/prof_copy -name '*kernel_trace.csv' | head -1)
gzip -c "$kt" > $out/pcie_kernel_trace.csv.gz; gzip -c "$mc" > $out/pcie_memory_copy_trace.csv.gz; ls -la $out | tail -30
