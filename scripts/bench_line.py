"""one-line summary of a bench.py JSON line on stdin (GPU probes): tag value ms_per_step stage times"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], d["ms_per_step"], d["config"].get("stage_ms_per_step"), d.get("roofline", {}).get("avg_launch_us"), "in-call", d["config"].get("ms_per_step_inside_step_call"), flush=True)
