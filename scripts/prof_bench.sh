#!/bin/bash
# rocprofv3 kernel stats of a bench.py run: scripts/prof_bench.sh <tag> [bench args...]   (PROF_STEPS / PROF_WARMUP: iterations traced)
export ISK_BENCH_NO_PMC=1     # (bench.py would otherwise run its own rocprofv3 counter passes for roofline.traffic)
tag=$1; shift
R=$(pwd); mkdir -p $R/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/bench.py --steps ${PROF_STEPS:-30} --warmup ${PROF_WARMUP:-5} --no-cpu-baseline --sustained-s 0 --no-f32-reference --spinup-s 0 "$@" > $R/gpurun_out/prof_$tag/log.txt 2>&1
cd $R
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(r["Name"][:60].ljust(60), "calls", r["Calls"].rjust(6), "avg_us", str(round(float(r["AverageNs"])/1e3,1)).rjust(8), "pct", r["Percentage"])
PY
tail -1 gpurun_out/prof_$tag/log.txt | cut -c1-300
