#!/bin/bash
# round 6: why is the live block slower inside the default bench process than as a process of its own?
set -u
root=$(pwd); out=$root/gpurun_out/r06_live; mkdir -p $out
export TMPDIR=/tmp
for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python bench.py --mode live --steps 100 --warmup 10 --no-cpu-baseline --no-live-nodes > $out/live_q$q.json 2> $out/live_q$q.err
  python -c "
import json; d=json.load(open('$out/live_q$q.json')); print('standalone GPU_MAX_HW_QUEUES=$q', d['value'], d['config']['stage_ms_per_sweep'])"
done
unset GPU_MAX_HW_QUEUES
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 1 > $out/inproc.json 2> $out/inproc.err
python - <<PY
import json
d = json.loads(open('$out/inproc.json').read().strip().splitlines()[-1])
for k in ('live_vlp16', 'live_hdl32'):
    b = d[k]; print('in-process', k, b.get('value'), (b.get('config') or {}).get('stage_ms_per_sweep'), b.get('value_nodes_concurrent'), b.get('error'))
print(d['config']['bench_phase_seconds'])
PY
LOAMX_BENCH_NO_BIND=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --repeat 1 > $out/inproc_nobind.json 2> $out/inproc_nobind.err
python - <<PY
import json
d = json.loads(open('$out/inproc_nobind.json').read().strip().splitlines()[-1])
for k in ('live_vlp16', 'live_hdl32'):
    b = d[k]; print('in-process, not bound to the NUMA node', k, b.get('value'), (b.get('config') or {}).get('stage_ms_per_sweep'), b.get('value_nodes_concurrent'))
print('value', d['value'])
PY
