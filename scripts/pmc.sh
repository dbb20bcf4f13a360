#!/bin/bash
# PMC passes for a bench workload (run on the GPU box): scripts/pmc.sh [tag [bench.py args...]]  -> gpurun_out/pmc_summary[_tag].json
#   scripts/pmc.sh                                                         the C3 bench
#   scripts/pmc.sh c5 --workload c5shard --forward-mode 2 --train-bf16     the C5 shard on its bf16 path
# separate passes, --kernel-trace only (never combined with other trace domains)
export ISK_BENCH_NO_PMC=1     # (bench.py would otherwise run its own rocprofv3 counter passes for roofline.traffic)
TAG=$1; [ $# -gt 0 ] && shift
BARGS="$*"
SUF=${TAG:+_$TAG}
R=$(pwd); P=$R/gpurun_out/pmc$SUF; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$n -o $n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 --sustained-s 0 --no-f32-reference --spinup-s 0 $BARGS > /dev/null 2> $P/$n.err || { tail -5 $P/$n.err; return 1; }
}
run fetch FETCH_SIZE && run write WRITE_SIZE && run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
cd $R && python3 scripts/pmc_summarize.py $P "$BARGS" > gpurun_out/pmc_summary$SUF.json && python3 - <<PY
import json
d=json.load(open("gpurun_out/pmc_summary$SUF.json"))
print(d["dominant_kernel"], d["mlp_fwd_traffic_bytes_per_launch"], d["dominant_kernel_derived"])
PY
