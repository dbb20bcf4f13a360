#!/bin/bash
# the documented environment knobs still give a correct engine: a quick subset of the GPU parity tests under each
# (run on the GPU box): scripts/knob_sanity.sh
export ISK_BENCH_NO_PMC=1     # (bench.py would otherwise run its own rocprofv3 counter passes for roofline.traffic)
export ISK_TEST_FWD_MODES=3
T="tests/test_gpu_parity.py -k run_"
for k in "ISK_FEAT_PF=2" "ISK_OVERLAP_FEATURIZE=0" "ISK_OVERLAP_FEATURIZE=1" "ISK_TRAIN_GRAPH=0" "ISK_SIDE_MIN_STEPS=0" "ISK_FEAT_EARLY_PCT=30" "ISK_FEAT_WG_PER_CU=2" "ISK_FEAT_BG_WGS=64"; do
  r=$(env $k timeout -k 10 300 python -m pytest $T -x -q -m gpu 2>&1 | tail -1)
  echo "$k | $r"
done
