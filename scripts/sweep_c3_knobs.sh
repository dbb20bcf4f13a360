#!/bin/bash
# same-box sweep of the side-stream / featurizer knobs on the C3 bench (boxes of the pool differ by 3-7 %: compare within one call)
export ISK_BENCH_NO_PMC=1     # (bench.py would otherwise run its own rocprofv3 counter passes for roofline.traffic)
source scripts/benchvar.sh
for rep in 1 2; do
  run "ISK_NOP=1" ""
  run "ISK_FEAT_BG_WGS=224" ""
  run "ISK_FEAT_BG_WGS=192" ""
  run "ISK_FEAT_PF=2" ""
  run "ISK_FEAT_WG_PER_CU=2" ""
  run "ISK_SIDE_EARLY=0" ""
  run "ISK_GRAD1_NW8=0" ""
  run "ISK_SIDE_MIN_STEPS=2" ""
done
