"""One iteration of the default bench from a rocprofv3 kernel trace (scripts/prof_bench.sh <tag>): every kernel that is not a training
kernel with its start (us after the forward's start) and duration, the training chain as one line, the gaps between the phases.
    python scripts/timeline_iter.py gpurun_out/prof_<tag>/<tag>_kernel_trace.csv [iteration]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r["Queue_Id"]) for r in rows)
fw = [i for i, e in enumerate(ev) if "mlp_fwd" in e[2] or "wide_fwd" in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(fw) // 2
i0, i1 = fw[k], fw[k + 1]
t0 = ev[i0][0]
print(f"iteration {k}: {(ev[i1][0] - t0) / 1e3:.1f} us from forward start to forward start (under the profiler)")
is_train = lambda n: "train_" in n or "opt_step" in n or "gemm16" in n or "opt_prep" in n or "head_kernel" in n
tr = [e for e in ev[i0:i1] if is_train(e[2])]
prev_end = None
for e in ev[i0:i1 + 1]:
    if is_train(e[2]):
        continue
    if tr and e[0] > tr[0][0] and prev_end is not None and prev_end < tr[0][0]:
        print(f"  {(tr[0][0] - t0) / 1e3:9.1f} +{(tr[-1][1] - tr[0][0]) / 1e3:7.1f}  q{tr[0][3]} training chain: {len(tr)} kernels, sum of durations {sum(x[1] - x[0] for x in tr) / 1e3:.1f} us"
              f" (gap to the kernel before it: {(tr[0][0] - prev_end) / 1e3:.1f} us)")
        prev_end = tr[-1][1]
    gap = "" if prev_end is None or e[3] != ev[i0][3] else f"  (gap {(e[0] - prev_end) / 1e3:.1f})"
    print(f"  {(e[0] - t0) / 1e3:9.1f} +{(e[1] - e[0]) / 1e3:7.1f}  q{e[3]} {e[2]}{gap}")
    if e[3] == ev[i0][3]:
        prev_end = e[1]
