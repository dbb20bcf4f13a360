#!/bin/bash
# A/B of the one-XCD small-batch chain (train_persist=2, experimental build) against the launch chain at --minibatch 100
set -o pipefail
out=gpurun_out/r5; mkdir -p $out
export ISK_LIB=$PWD/devlib/libisokann_hip_exp.so ISK_BENCH_NO_PMC=1
for i in 1 2; do
  for m in 0 2; do
    ISK_TRAIN_PERSIST=$m timeout -k 10 200 python bench.py --minibatch 100 --opt nesterov --no-cpu-baseline ${BENCH_EXTRA} > $out/chain_b${m}_$i.json 2> $out/chain_b${m}_$i.err || { tail -5 $out/chain_b${m}_$i.err; exit 1; }
  done
done
python - <<PY
import json
for n in ("b0_1","b2_1","b0_2","b2_2"):
    l=json.loads(open("$out/chain_%s.json"%n).read().strip().splitlines()[-1])
    print(n, l["ms_per_step"], l["value"], l.get("config"), l.get("roofline",{}).get("step"))
PY
