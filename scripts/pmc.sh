#!/bin/bash
# PMC passes for the C3 bench (run on the GPU box): scripts/pmc.sh  -> gpurun_out/pmc_summary.json
# separate passes, --kernel-trace only (never combined with other trace domains)
R=$(pwd); mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc/$n -o $n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 > /dev/null 2> $R/gpurun_out/pmc/$n.err || { tail -5 $R/gpurun_out/pmc/$n.err; return 1; }
}
run fetch FETCH_SIZE && run write WRITE_SIZE && run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
cd $R && python3 scripts/pmc_summarize.py gpurun_out/pmc > gpurun_out/pmc_summary.json && python3 - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_summary.json"))
print(d["dominant_kernel"], d["mlp_fwd_traffic_bytes_per_launch"], d["dominant_kernel_derived"])
PY
