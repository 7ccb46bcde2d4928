#!/bin/bash
# install_evidence.sh <tag> [round]: copy what scripts/gpu_evidence.sh <tag> left under gpurun_out/ into profiles/ (tracked) under the
# round's names and regenerate BASELINE.md's table from the bench line.
set -eu
tag=$1; r=${2:-r06}
root=$(cd "$(dirname "$0")/.." && pwd); g=$root/gpurun_out; p=$root/profiles
cpif() { if [ -s "$1" ]; then cp "$1" "$2"; else echo "missing: $1"; fi; }
tail -n 1 $g/$tag/bench.json > $p/bench_$r.json
cpif $g/$tag/gpu_tests.log $p/${r}_gpu_tests.log
cpif $g/$tag/kernel_stats.csv $p/${r}_bench_kernel_stats.csv
cpif $g/$tag/kernels.txt $p/${r}_bench_kernels_per_step.txt
cpif $g/$tag/chain.txt $p/${r}_bench_chain_timeline.txt
cpif $g/$tag/chain_by_registration.txt $p/${r}_bench_chain_timeline_by_registration.txt
cpif $g/$tag/pmc_summary.json $p/${r}_pmc_summary.json
cpif $g/$tag/sequential_summary.txt $p/${r}_sequential_kernel_durations.txt
cpif $g/$tag/live_vlp16_pmc_summary.json $p/${r}_live_vlp16_pmc_summary.json
cpif $g/$tag/live_hdl32_pmc_summary.json $p/${r}_live_hdl32_pmc_summary.json
cpif $g/${tag}_live1/kernel_stats.csv $p/${r}_live_vlp16_kernel_stats.csv
cpif $g/${tag}_live1/summary.txt $p/${r}_live_vlp16_kernels_per_step.txt
cpif $g/${tag}_live2/kernel_stats.csv $p/${r}_live_hdl32_kernel_stats.csv
cpif $g/${tag}_live2/summary.txt $p/${r}_live_hdl32_kernels_per_step.txt
tail -n 1 $g/${tag}_live1/bench.json > $p/bench_${r}_live_vlp16_standalone.json
tail -n 1 $g/${tag}_live2/bench.json > $p/bench_${r}_live_hdl32_standalone.json
python - "$g/$tag" "$p/${r}_stream_sweep.json" "$p/bench_$r.json" <<'PY'
import json, sys
src, dst, line = sys.argv[1:4]
old = json.load(open(dst)) if __import__("os").path.exists(dst) else {}
d = json.load(open(line))
keys = ("value", "value_median", "ms_per_step")
old["8"] = {**{k: d.get(k) for k in keys}, "stage_ms_per_step": d["config"].get("stage_ms_per_step"), "path_hbm_frac": d["summary"].get("path_hbm_frac"),
            "path_algorithmic_bytes_per_sweep": d["config"].get("path_algorithmic_bytes_per_sweep", old.get("8", {}).get("path_algorithmic_bytes_per_sweep"))}
for s in (1, 4, 16, 32):
    try:
        x = json.loads(open(f"{src}/streams_{s}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print("no stream figure for", s, e); continue
    old[str(s)] = {**{k: x.get(k) for k in keys}, "stage_ms_per_step": x["config"].get("stage_ms_per_step"), "path_hbm_frac": (x.get("summary") or {}).get("path_hbm_frac"),
                   "path_algorithmic_bytes_per_sweep": x["config"].get("path_algorithmic_bytes_per_sweep")}
json.dump(old, open(dst, "w"), indent=1)
PY
# unprofiled host trace: keep the earlier runs of the file's header, put this run first
if [ -s $g/$tag/pipe_trace_summary.txt ]; then
  { echo "# unprofiled host trace of the batched window, final code (scripts/gpu_trace_session.sh; evidence call $tag)"; cat $g/$tag/pipe_trace_summary.txt; } > $p/${r}_pipe_trace_summary.txt
fi
python - $root <<'PY'
import subprocess, sys, re
root = sys.argv[1]
tab = subprocess.run([sys.executable, "scripts/baseline_table.py", "profiles/bench_r06.json", "profiles/r06_stream_sweep.json"], capture_output=True, text=True, check=True, cwd=root).stdout
s = open(f"{root}/BASELINE.md").read()
a, b = s.index("<!-- TABLE BEGIN -->") + len("<!-- TABLE BEGIN -->"), s.index("<!-- TABLE END -->")
open(f"{root}/BASELINE.md", "w").write(s[:a] + "\n" + tab.strip() + "\n" + s[b:])
print("BASELINE.md table regenerated")
PY
