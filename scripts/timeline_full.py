#!/usr/bin/env python3
"""Every kernel of one step of a rocprofv3 kernel trace: scripts/timeline_full.py <kernel_trace.csv> [step_index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fw = [i for i, r in enumerate(rows) if "mlp_fwd_bf16x3p" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(fw) // 2
a, b = fw[k], fw[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = {}
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Queue_Id", "?")
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    n = r["Kernel_Name"].replace("void ", "")[:40]
    print(f"q{q:>3} {s/1e3:9.1f} -> {e/1e3:9.1f} us  ({(e-s)/1e3:7.1f})  gap {gap/1e3:6.1f}  {n}")
print("step", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us")
