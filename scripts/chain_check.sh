#!/bin/bash
# tests of the one-XCD chain (experimental build), then its stamp table at --minibatch 100
set -o pipefail
mkdir -p gpurun_out/r5
export ISK_LIB=$PWD/devlib/libisokann_hip_exp.so
ISK_TEST_PERSIST_MODE=2 timeout -k 10 300 python -m pytest tests/test_gpu_epoch_persist.py -x -q > gpurun_out/r5/chain_t.txt 2>&1 || { tail -30 gpurun_out/r5/chain_t.txt; exit 1; }
tail -2 gpurun_out/r5/chain_t.txt
ISK_BENCH_NO_PMC=1 ISK_TRAIN_PERSIST=2 ISK_DEBUG_STAMPS=1 timeout -k 10 200 python bench.py --minibatch 100 --opt ${OPT:-nesterov} --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/r5/chain_stamps.txt 2>&1 || { tail -5 gpurun_out/r5/chain_stamps.txt; exit 1; }
grep -a "stamps" gpurun_out/r5/chain_stamps.txt | tail -4
grep -a '"metric"' gpurun_out/r5/chain_stamps.txt | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', l['ms_per_step'])"
