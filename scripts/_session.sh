#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s9
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s9/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s9/tests.log
tail -5 gpurun_out/s9/tests.log
for r in 1 2; do
  timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > gpurun_out/s9/live_vlp_$r.json 2>gpurun_out/s9/err_vlp_$r.log
  timeout 300 python bench.py --mode live --sensor HDL-32 --map-points 500000 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > gpurun_out/s9/live_hdl_$r.json 2>gpurun_out/s9/err_hdl_$r.log
done
LOAMX_MAP_TRACE=1 timeout 300 python bench.py --mode live --steps 64 --warmup 10 --no-cpu-baseline --no-live-nodes 2>&1 >/dev/null | grep "map trace" | tail -4
for f in gpurun_out/s9/live_*.json; do echo "$f $(grep -h -o '"value": [0-9.]*\|"value_nodes_concurrent": [0-9.]*\|"sweeps_per_s": [0-9.]*\|"stage_ms_per_sweep": {[^}]*}' $f | tr '\n' ' ')"; done
