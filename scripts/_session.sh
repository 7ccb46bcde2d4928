#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/s5
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s5/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s5/tests.log
tail -5 gpurun_out/s5/tests.log
for r in 1 2; do
  timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > gpurun_out/s5/live_vlp_$r.json 2>gpurun_out/s5/err_vlp_$r.log
  timeout 300 python bench.py --mode live --sensor HDL-32 --map-points 500000 --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > gpurun_out/s5/live_hdl_$r.json 2>gpurun_out/s5/err_hdl_$r.log
done
timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/s5/live_vlp_nodes.json 2>gpurun_out/s5/err_vlp_nodes.log
tail -3 gpurun_out/s5/err_vlp_1.log
for f in gpurun_out/s5/live_*.json; do echo "$f $(grep -h -o '"value": [0-9.]*\|"value_nodes_concurrent": [0-9.]*\|"sweeps_per_s": [0-9.]*\|"stage_ms_per_sweep": {[^}]*}' $f | tr '\n' ' ')"; done
