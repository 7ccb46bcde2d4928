"""BASELINE.md section 3 from a bench line: python scripts/baseline_table.py profiles/bench_r06.json [profiles/r06_stream_sweep.json]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sweep = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else {}
src = sys.argv[1]


def pe(blk):
    p = blk.get("pose_err_vs_oracle") or {}
    mp = p.get("mapped_pose") or {}
    return p, mp


cb = d.get("cpu_baseline", {})
rows = []
rows.append("| Config (`BASELINE.json`) | CPU oracle sweeps/s, one core (same box, same sweeps) | GPU sweeps/s, one MI355X | ms | dominant kernel: achieved GB/s, fraction of 8 TB/s | pose max |difference| vs the oracle chain (m) / bar | where |")
rows.append("|---|---|---|---|---|---|---|")
rows.append("| 0. VLP-16, 100 k map, CPU only | the oracle itself (row 1's CPU leg is the same code) | n/a | n/a | n/a | defines truth (`tests/test_oracle_pipeline.py`, `test_ref_pinning.py`) | — |")
for key, name in (("live_vlp16", "1. VLP-16, 200 k LIVE map, one sweep in flight (sequential SLAM)"), ("live_hdl32", "2. HDL-32, 500 k LIVE map, one sweep in flight")):
    b = d.get(key) or {}
    if "value" not in b:
        continue
    p, mp = pe(b)
    r = b.get("roofline", {})
    rows.append(f"| {name} | {b.get('cpu_baseline', {}).get('value')} | **{b['value']:,.0f}** (three concurrent nodes over host messages: {b.get('value_nodes_concurrent')}; host-message chain "
                f"{b['config']['host_message_chain']['sweeps_per_s']:,.0f}) | {b['ms_per_step']} per sweep | `k_gn_iter` {r.get('achieved')} GB/s, {r.get('frac')} (one sweep cannot fill the device: latency) | "
                f"{mp.get('max_m', float('nan')):.2e} / {p.get('bar_free_running', float('nan')):.2e} (free running over a live map; envelope {((p.get('reference_envelope') or {}).get('max_m') or float('nan')):.2e}) | `{src}` → `{key}` |")
p, mp = pe(d)
r = d["roofline"]
vl = d.get("value_long") or {}
pl, mpl = pe(vl)
rows.append(f"| 3. HDL-64E, 1 M frozen map, batch 32 over 4 GPUs = 8 streams per GPU (**the metric's configuration**) | {cb.get('value')} (3-stage pipeline, 3 cores: {cb.get('pipelined_value')}) | "
            f"**{d['value']:,.0f}** (median of {d.get('value_repeats')} windows {d.get('value_median'):,.0f}; 400-step window {vl.get('value', float('nan')):,.0f}; PCIe-inclusive {(d.get('pcie_inclusive') or {}).get('value')}"
            + (f"; streams per GPU 16 / 32: {sweep['16']['value']:,.0f} / {sweep['32']['value']:,.0f}" if sweep else "") + ") | "
            f"{d['ms_per_step']} per 8-sweep step | `{r['kernel'].split('::')[-1]}` {r['achieved']} GB/s, {r['frac']} ({r.get('latency_model', {}).get('us_per_iteration')} µs per iteration, floor 4.5); "
            f"`k_gn_iter` {(r.get('k_gn_iter') or {}).get('frac')}; whole path {d['config']['path_hbm_frac']} | "
            f"{mp.get('max_m', float('nan')):.2e} / 1e-4 (23 sweeps); 405 sweeps: {mpl.get('max_m', float('nan')):.2e} / {pl.get('bar_free_running', float('nan')):.2e}, "
            f"per step from identical state {(pl.get('per_step_from_identical_state') or {}).get('max_m', float('nan')):.2e} / 1e-4 | `{src}` |")
b = d.get("map_2m") or {}
if "value" in b:
    p, mp = pe(b)
    r = b.get("roofline", {})
    rows.append(f"| 4. HDL-64E, 2 M frozen map, batch 64 over 8 GPUs — **one-GPU point only**; no scaling curve has ever been measured (N > 1 never ran on hardware) | "
                f"{b.get('cpu_baseline', {}).get('value')} | **{b['value']:,.0f}** (median of {b.get('value_repeats')}: {b.get('value_median'):,.0f}) | {b['ms_per_step']} per 8-sweep step | "
                f"`k_gn_iter` {r.get('achieved')} GB/s, {r.get('frac')} | {mp.get('max_m', float('nan')):.2e} / 1e-4 | `{src}` → `map_2m` |")
print("\n".join(rows))
