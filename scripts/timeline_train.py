#!/usr/bin/env python3
"""One optimizer step of a rocprofv3 kernel trace, kernel by kernel: scripts/timeline_train.py <kernel_trace.csv> <first-kernel-substring> [occurrence]
(developer tool: prints the kernels from the given occurrence of the step's first kernel to the next one, with gaps)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = {}
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Queue_Id", "?")
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    n = r["Kernel_Name"].replace("void ", "")[:50]
    g = "x".join(str(int(r.get(k_, 0) or 0)) for k_ in ("Grid_Size_X", "Grid_Size_Y") if k_ in r)
    print(f"q{q:>3} {s/1e3:9.1f} -> {e/1e3:9.1f} us  ({(e-s)/1e3:7.1f})  gap {gap/1e3:6.1f}  grid {g:>12}  {n}")
print("step", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us")
