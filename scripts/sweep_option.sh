#!/bin/bash
# sweep one engine option on a bench workload: scripts/sweep_option.sh ISK_<KEY> "v1 v2 ..." [bench args...]
set -o pipefail
key=$1; vals=$2; shift 2
export ISK_BENCH_NO_PMC=1
for rep in 1 2; do
  for v in $vals; do
    env $key=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-f32-reference --sustained-s 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$key', '$v', 'rep', $rep, 'ms', round(d['ms_per_step'],4), {k:(round(x,3) if x else x) for k,x in r['phases_ms'].items()})
" || exit 1
  done
done
