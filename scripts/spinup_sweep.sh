#!/bin/bash
# the driver's window (--steps 20 --warmup 5) under different spin-up lengths
export ISK_BENCH_NO_PMC=1
mkdir -p gpurun_out/r5
for rep in 1 2 3; do
  for s in 0.3 1.0 2.0; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-reference --sustained-s 0 --spinup-s $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spinup', $s, 'rep', $rep, 'ms_per_step', round(d['ms_per_step'],4))
" || exit 1
  done
done
