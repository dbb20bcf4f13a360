"""per-kernel summary of a rocprofv3 --kernel-trace run (its results .db, or the *_kernel_stats.csv): python scripts/kstats.py <file> [top]"""
import csv
import sqlite3
import sys

path, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
if path.endswith(".csv"):
    for r in list(csv.DictReader(open(path)))[:top]:
        print(f"{r['Name'][:100]:100s} n={int(r['Calls']):6d} avg_us={float(r['AverageNs']) / 1e3:9.1f} total_ms={float(r['TotalDurationNs']) / 1e6:9.2f} {r['Percentage']}%")
else:
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id "
         f"group by s.kernel_name order by 4 desc limit {top}")
    tot = c.execute(f"select sum(end-start)/1e6 from {kd}").fetchone()[0]
    for r in c.execute(q):
        print(f"{r[0][:100]:100s} n={r[1]:6d} avg_us={r[2]:9.1f} total_ms={r[3]:9.2f} {100 * r[3] / tot:5.1f}%")
