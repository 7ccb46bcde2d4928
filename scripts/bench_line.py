"""Condensed view of bench.py's JSON line (stdin)."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(" ".join(sys.argv[1:]), "streams", d["config"]["streams_per_gpu"], "handles", d["config"]["handles_per_gpu"], "value", d["value"],
      "ms/step", d["ms_per_step"], d["config"]["stage_ms_per_step"])
