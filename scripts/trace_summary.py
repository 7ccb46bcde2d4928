#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per-kernel duration statistics and the kernel sequence of the last full step per stream."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
print("kernel n avg med p90 max total(us)")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k:60s} n={len(v):5d} avg={sum(v)/len(v):7.1f} med={v2[len(v)//2]:7.1f} p90={v2[int(len(v)*0.9)]:7.1f} max={v2[-1]:7.1f} total={sum(v):10.1f}")
if len(sys.argv) > 2:
    n = int(sys.argv[2])
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    tail = rows[-n:]
    t0 = int(tail[0]["Start_Timestamp"])
    for r in tail:
        print(f'{(int(r["Start_Timestamp"])-t0)/1000:9.1f} +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000:7.1f}  q{r["Queue_Id"]:>3s} grid={r["Grid_Size_X"]:>7s}x{r["Grid_Size_Y"]:>3s}  {r["Kernel_Name"].split("(")[0][:50]}')
