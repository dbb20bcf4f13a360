#!/bin/bash
# regenerate the committed evidence under profiles/ (run on the GPU box): scripts/refresh_profiles.sh r01
# (a gpurun call is limited to 20 minutes: SKIP_PROF=1 runs the bench variants only, SKIP_BENCH=1 the traces and counter passes only)
tag=${1:-r05}
mkdir -p gpurun_out/ref
# no `set -e` (one failed trace must not stop the others), but every failed step is NAMED at the end and the script fails: a
# stale or missing file under profiles/ is then visible instead of silent
FAILED=""
export ISK_BENCH_NO_PMC=1     # only the headline run (b c3) measures roofline.traffic itself: two more rocprofv3 passes per bench run otherwise
set -o pipefail            # a bench that dies in front of `| tail -1` must count as failed
fail() { FAILED="$FAILED $1"; echo "[refresh_profiles] FAILED: $1" >&2; }
b() { name=$1; shift; python3 bench.py "$@" 2> gpurun_out/ref/$name.err | tail -1 > gpurun_out/ref/${tag}_bench_$name.json || fail bench_$name; python3 -c "
import json,sys; d=json.load(open('gpurun_out/ref/${tag}_bench_$name.json')); print('$name', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,1), 'M/s frac', round(d['roofline']['frac'],3))" || fail bench_${name}_line; }
if [ -z "$SKIP_BENCH" ]; then
ISK_BENCH_NO_PMC= b c3
b c3_bf16x3fwd --forward-mode 1
b c3_f32fwd --forward-mode 0
b c3_cached --cached-features
b c3_mb0 --minibatch 0
b c3_mb100_nesterov --minibatch 100 --opt nesterov
b c2 --workload c2
b c5shard --workload c5shard --forward-mode 1 --no-cpu-baseline                               # split-bf16 forward (f32-equivalent), f32 training
b c5shard_f32fwd --workload c5shard --forward-mode 0 --no-cpu-baseline
b c5shard_bf16fwd --workload c5shard --forward-mode 2 --no-train-bf16 --no-cpu-baseline      # fused bf16 forward, f32 training: the in-house NT GEMMs in Float32 arithmetic (default since round 4)
b c5shard_bf16 --workload c5shard --forward-mode 2 --train-bf16 --no-cpu-baseline          # the arithmetic BASELINE names for C5: forward AND training GEMMs on bf16
# the same through the native data-parallel driver with every collective on a one-rank communicator: what ONE of C5's 8 ranks runs
ISK_DP_FORCE_COLL=1 b c5shard_bf16_dp1 --workload c5shard --forward-mode 2 --train-bf16 --force-dp --no-cpu-baseline
b c3_bf16fwd --forward-mode 2 --no-cpu-baseline
b c3_pullback --pullback 2048 --no-cpu-baseline                                          # + the coordinate pullback on 2048 columns (field "pullback")
# (the A/B runs of train_persist / fuse_tail / fuse_opt_rows / g16_waves / epoch_gather are round-4 evidence, profiles/r04_*: those
# variants exist in -DISK_EXPERIMENTAL builds only since round 5)
ISK_FUSE_OPT=0 b c3_mb100_nesterov_nofuseopt --minibatch 100 --opt nesterov --no-cpu-baseline
ISK_FUSE_HEAD=0 b c5shard_bf16_nohead --workload c5shard --forward-mode 2 --train-bf16 --no-cpu-baseline          # three launches instead of the head kernel
ISK_FUSE_BOUNDARY=0 b c5shard_bf16_noboundary --workload c5shard --forward-mode 2 --train-bf16 --no-cpu-baseline  # opt_step + prep_images instead of the boundary launch
ISK_TAIL_NARROW=0 b c3_generic_tail --no-cpu-baseline                                    # round 3's train_tail kernel, for the A/B on this box
# one rank's share of the 8-way strong-scaled runs (C4: N = 1250, local batch 125; collectives on a one-rank communicator)
b c4_shard_of_8 --shard-of 8 --no-cpu-baseline
b c4_shard_of_4 --shard-of 4 --no-cpu-baseline
b c4_shard_of_2 --shard-of 2 --no-cpu-baseline
fi
if [ -z "$SKIP_PROF" ]; then
bash scripts/prof_bench.sh c3stats > gpurun_out/ref/prof.txt 2>&1 || fail "bash_prof_bench.sh_c3stats_prof.txt"
cp gpurun_out/prof_c3stats/c3stats_kernel_stats.csv gpurun_out/ref/${tag}_kernel_stats_bench_c3.csv || fail "__tag__kernel_stats_bench_c3.csv"
# The epoch graph of this configuration has 400 kernel nodes.  rocprofv3 (ROCm 7.2) segfaults INSIDE its interception of
# hipGraphLaunch once about 10 000 graph-launched kernels have been traced in one process (SIGSEGV at a page-aligned address,
# frames in the tool library under train_epoch -> hipGraphLaunch; 12 iterations = 4 800 kernels trace fine, 35 = 14 000 crash;
# without the profiler, and with the profiler on eager launches, any length runs): a capacity limit of the tool, not the engine's
# capture.  So the graph path is traced over 12 iterations (round 3 traced eager launches instead).
rm -rf gpurun_out/prof_c3mb100
PROF_STEPS=10 PROF_WARMUP=2 bash scripts/prof_bench.sh c3mb100 --minibatch 100 --opt nesterov >> gpurun_out/ref/prof.txt 2>&1 || fail "PROF_STEPS10_PROF_WARMUP2_bash_prof_bench.sh"
cp gpurun_out/prof_c3mb100/c3mb100_kernel_stats.csv gpurun_out/ref/${tag}_kernel_stats_bench_c3_mb100_nesterov.csv || fail "cp_c3mb100_kernel_stats.csv_tag_kernel_stats_ben"
bash scripts/prof_bench.sh c5stats --workload c5shard --forward-mode 2 --train-bf16 >> gpurun_out/ref/prof.txt 2>&1 || fail "bash_prof_bench.sh_c5stats_--workload"
cp gpurun_out/prof_c5stats/c5stats_kernel_stats.csv gpurun_out/ref/${tag}_kernel_stats_bench_c5shard.csv || fail "__tag__kernel_stats_bench_c5shard.csv"
bash scripts/prof_bench.sh c5f32 --workload c5shard --forward-mode 2 --no-train-bf16 >> gpurun_out/ref/prof.txt 2>&1 || fail "bash_prof_bench.sh_c5f32_--workload"
cp gpurun_out/prof_c5f32/c5f32_kernel_stats.csv gpurun_out/ref/${tag}_kernel_stats_bench_c5shard_f32train.csv || fail "__tag__kernel_stats_bench_c5shard_f32train.csv"
bash scripts/prof_bench.sh c4s8 --shard-of 8 >> gpurun_out/ref/prof.txt 2>&1 || fail "bash_prof_bench.sh_c4s8_--shard-of"
cp gpurun_out/prof_c4s8/c4s8_kernel_stats.csv gpurun_out/ref/${tag}_kernel_stats_bench_c4_shard_of_8.csv || fail "__tag__kernel_stats_bench_c4_shard_of_8.csv"
bash scripts/pmc.sh > gpurun_out/ref/pmc.txt 2>&1 || fail "bash_pmc.sh_pmc.txt_21"
python3 - <<PY
import json
d = json.load(open("gpurun_out/pmc_summary.json"))
d["mlp_fwd_traffic_bytes_per_launch"] = {"f16x2": d["mlp_fwd_traffic_bytes_per_launch"]}     # bench.py looks the forward mode up
json.dump(d, open("gpurun_out/ref/${tag}_pmc_summary.json", "w"), indent=1)
PY
tail -3 gpurun_out/ref/pmc.txt
# the same passes on the C5 shard's bf16 path (dominant kernel there: the fused wide forward)
bash scripts/pmc.sh c5 --workload c5shard --forward-mode 2 --train-bf16 > gpurun_out/ref/pmc_c5.txt 2>&1 || fail "bash_pmc.sh_c5_--workload"
cp gpurun_out/pmc_summary_c5.json gpurun_out/ref/${tag}_pmc_summary_c5.json || fail "__tag__pmc_summary_c5.json"
tail -2 gpurun_out/ref/pmc_c5.txt
# L2 read-request split (TCC_EA0_RDREQ by size, hit / miss) of the forward kernels
bash scripts/pmc_tcc.sh c3 > gpurun_out/ref/tcc_c3.txt 2>&1 || fail "bash_pmc_tcc.sh_c3_tcc_c3.txt"
bash scripts/pmc_tcc.sh c5 --workload c5shard --forward-mode 2 --train-bf16 > gpurun_out/ref/tcc_c5.txt 2>&1 || fail "bash_pmc_tcc.sh_c5_--workload"
cp gpurun_out/pmc_tcc_c3.json gpurun_out/ref/${tag}_pmc_tcc_c3_raw.json 2>/dev/null || fail "cp_pmc_tcc_c3.json_tag_pmc_tcc_c3_raw.json_null"
cp gpurun_out/pmc_tcc_c5.json gpurun_out/ref/${tag}_pmc_tcc_c5_raw.json 2>/dev/null || fail "cp_pmc_tcc_c5.json_tag_pmc_tcc_c5_raw.json_null"
python3 scripts/c3sub_table.py > gpurun_out/ref/${tag}_c3sub_error_table.md 2> gpurun_out/ref/c3sub_table.err || fail "python3_c3sub_table.py_tag_c3sub_error_table.md_"
fi   # SKIP_PROF
if [ -n "$FAILED" ]; then echo "[refresh_profiles] steps that FAILED (their files under gpurun_out/ref are missing or stale):$FAILED" >&2; exit 1; fi
echo "[refresh_profiles] all steps ok"
