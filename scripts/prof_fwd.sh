#!/bin/bash
# kernel-level timing of scripts/time_fwd.py under rocprofv3: scripts/prof_fwd.sh <mode> [tag]
export ISK_BENCH_NO_PMC=1     # (bench.py would otherwise run its own rocprofv3 counter passes for roofline.traffic)
mode=${1:-1}; tag=${2:-fwd}
R=$(pwd); mkdir -p $R/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/scripts/time_fwd.py $mode > $R/gpurun_out/prof_$tag/log.txt 2>&1
cd $R
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(r["Name"][:70], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,1), "pct", r["Percentage"])
PY
