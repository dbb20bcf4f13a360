#!/bin/bash
# ms per step of the default bench as a function of the timed window's length K: t(K) = a + c / K separates a per-window cost c
export ISK_BENCH_NO_PMC=1
for rep in 1 2; do
  for k in 5 10 20 40 80 160 640; do
    timeout -k 10 200 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-f32-reference --sustained-s 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K', $k, 'rep', $rep, 'ms_per_step', round(d['ms_per_step'],4), 'window_ms', round(d['ms_per_step']*$k,3))
" || exit 1
  done
done
