#!/bin/bash
# A/B of one engine option on the default bench (C3): scripts/ab_option.sh ISK_<KEY> v0 v1 [bench args...]
set -o pipefail
key=$1; v0=$2; v1=$3; shift 3
out=gpurun_out/r5; mkdir -p $out
export ISK_BENCH_NO_PMC=1
for i in 1 2; do
  for v in $v0 $v1; do
    env $key=$v timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $out/ab_${key}_${v}_$i.json 2> $out/ab_${key}_${v}_$i.err || { tail -5 $out/ab_${key}_${v}_$i.err; exit 1; }
  done
done
python - <<PY
import json
for i in (1,2):
    for v in ("$v0","$v1"):
        l=json.loads(open("$out/ab_${key}_%s_%d.json"%(v,i)).read().strip().splitlines()[-1])
        r=l["roofline"]
        print("$key", v, i, "ms_per_step", round(l["ms_per_step"],4), "sustained", round(l["sustained"]["ms_per_step"],4) if l.get("sustained") else None, "phases", {k:(round(x,4) if x else x) for k,x in r["phases_ms"].items()})
PY
