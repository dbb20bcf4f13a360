#!/usr/bin/env python
"""bench.py -- ISOKANN train-step throughput on MI355X (BASELINE.json metric).

One "step" = one run!(iso, 1, 1) iteration (src/iso.jl:72-94 of the reference) on synthetic
data already resident in HBM: pairwise-distance featurization of xs and ys (per iteration,
as the north-star defines the hot path), MLP forward over the N*K Koopman samples (+ N start
points for ISA), K-mean, shift-scale / ISA target, one training epoch (weighted MSE fwd/bwd +
WeightDecay+Adam) on xs.   value = Koopman samples (N*K, summed over ranks) per second.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          (one rank per GPU, RCCL)

Multi-GPU is WEAK scaling: every rank holds its own C3-sized shard (N=10000 starts, K=20) and a
fixed per-rank minibatch, so the global batch is minibatch*N_gpus and the number of optimizer
steps per epoch stays constant; one flat [gradient|loss] all-reduce per step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def build_iso(pkg, wl, seed, device, args):
    from isokann_jl_amd import synth
    c = synth.WORKLOADS[wl]
    xs, ys = synth.synthetic_coords(c["atoms"], c["N"], c["K"], seed=seed)
    if c["feat"] == "all":
        feat, F = pkg.FeaturesAll(), c["atoms"] * (c["atoms"] - 1) // 2
    else:
        feat, F = pkg.FeaturesPairs(synth.random_pairs(c["atoms"], c["feat"], seed=0)), c["feat"]
    widths = c.get("widths") or pkg.pairnet_widths(F, nout=c["nout"])
    model = pkg.densenet(widths, layernorm=c.get("layernorm", True), seed=0)       # glorot init, identical on every rank
    opt = pkg.AdamRegularized() if args.opt == "adam" else pkg.NesterovRegularized()
    data = pkg.SimulationData(xs, ys, feat)
    iso = pkg.Iso(data, opt=opt, model=model, minibatch=args.minibatch, device=device,
                  featurize_every_iteration=not args.cached_features, seed=seed, forward_mode=args.forward_mode)
    del xs, ys
    return iso, c, widths


def algorithmic_flops(widths, n, k, nb_rows):
    """F_fwd = 2*sum(in*out) per sample"""
    ffwd = 2 * sum(a * b for a, b in zip(widths[:-1], widths[1:]))
    isa = 1 if widths[-1] > 1 else 0
    fwd_launch = (n * k + isa * n) * ffwd                       # the dominant kernel (one launch)
    step = fwd_launch + nb_rows * (3 * ffwd - 2 * widths[0] * widths[1])
    return ffwd, fwd_launch, step


def cpu_baseline(wl, args):
    """the CPU oracle (a port of the reference semantics, NumPy/BLAS fp32) timed on a bounded
    sample of the same workload on this host's cores"""
    from oracle import isokann_oracle as O
    from isokann_jl_amd import synth
    c = synth.WORKLOADS[wl]
    ns = min(c["N"], args.cpu_sample)
    xs, ys = synth.synthetic_coords(c["atoms"], ns, c["K"], seed=1)
    if c["feat"] == "all":
        feat = O.FeaturesAll()
        F = c["atoms"] * (c["atoms"] - 1) // 2
    else:
        feat = O.FeaturesPairs(synth.random_pairs(c["atoms"], c["feat"], seed=0))
        F = c["feat"]
    om = O.glorot_model(c.get("widths") or O.pairnet_widths(F, nout=c["nout"]), np.random.default_rng(0),
                        layernorm=c.get("layernorm", True), dtype=np.float32)
    rule = O.AdamRegularized() if args.opt == "adam" else O.NesterovRegularized()
    st = O.setup(rule, om)
    rng = np.random.default_rng(0)
    mb = min(args.minibatch, ns) if args.minibatch else 0

    def step():
        data = O.SimulationData(xs, ys, feat, dtype=np.float32)          # featurization is part of the step
        iso = O.Iso(data, om, st, minibatch=mb)
        O.run(iso, 1, 1, rng=rng)

    step()
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or (time.perf_counter() - t0 < 10.0 and reps < 50):
        step()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": ns * c["K"] / dt, "unit": "Koopman-samples/s", "cores": int(cores), "kind": "port",
            "sample": f"{ns} of {c['N']} start points x K={c['K']} (same featurizer/model/optimizer, fp32 NumPy+BLAS oracle, "
                      f"{reps} timed steps of {dt:.2f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c5shard"])
    ap.add_argument("--minibatch", type=int, default=1000, help="per-rank minibatch (scripts/villin.jl:37 uses 1000); 0 = full batch")
    ap.add_argument("--opt", default="adam", choices=["adam", "nesterov"])
    ap.add_argument("--cached-features", action="store_true", help="featurize once at set_data (reference semantics) instead of every iteration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1000)
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--forward-mode", type=int, default=3, choices=[0, 1, 2, 3],
                    help="0 = f32 MFMA forward, 1 = split-bf16 (bf16x3, f32-equivalent), 2 = plain bf16 operands (reduced precision), "
                         "3 = split-f16 (f16x2, f32-equivalent; planes written by the featurizer)")
    ap.add_argument("--force-dp", action="store_true", help="use the data-parallel driver even with one rank (rehearsal)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    ge.build()
    pkg = ge.load_package()
    if pkg.capi.device_count() <= local_rank:
        sys.exit("no HIP device for this rank: the product path has no CPU fallback")

    dist = None
    use_dp = world > 1 or args.force_dp
    real_stdout = None
    if use_dp:
        # RCCL prints a version banner on stdout; the driver wants exactly one JSON line there
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.forward_mode == 3 and args.workload == "c5shard":
        args.forward_mode = 1      # 512-wide layers run on the layered path, where a split-f16 request means the split-bf16 kernels
    iso, c, widths = build_iso(pkg, args.workload, seed=rank, device=local_rank, args=args)
    N, K = c["N"], c["K"]
    bs = N if (args.minibatch == 0 or N < args.minibatch) else args.minibatch
    nb_rows = (N // bs) * bs

    if not use_dp:
        def step(k=1):
            # run!(iso, k): k iterations inside the engine, as the reference's users call it -- one host round trip per
            # iteration for the loss (src/iso.jl:81), no Python between two iterations
            pkg.run(iso, k, 1)

        def sync():
            pkg.capi.check(iso._lib.isk_synchronize(iso._h))

        def barrier():
            pass
    else:
        import torch
        from isokann_jl_amd import dp
        comm = dp.Comm(device=torch.device("cuda", local_rank))
        local = dp.HipLocal(iso)
        rng = np.random.default_rng(1000 + rank)
        gmb = 0 if args.minibatch == 0 else args.minibatch * world       # fixed per-rank batch
        counts = dp.shard_counts(local, comm)

        def step(k=1):
            # rng=None: the engine draws the shuffle orders itself (seeded), hidden behind the forward, instead of ~0.15 ms of
            # host work between two iterations
            iso.losses.extend(dp.dp_run(local, comm, k, 1, gmb, None, counts=counts))

        def sync():
            torch.cuda.synchronize()

        def barrier():
            dist.barrier()

    watchdog = None
    if use_dp and world > 1:
        # a collective that a peer never joins blocks inside the engine for good (there is nothing to catch): leave with a
        # diagnostic instead of hanging until the launcher's own limit kills the job
        import threading

        def _bail():
            print(f"[bench] rank {rank}: no progress for {int(os.environ.get('ISK_BENCH_WATCHDOG_S', '420'))} s in the data-parallel run "
                  "(a collective is waiting for a peer?) -- aborting", file=sys.stderr, flush=True)
            os._exit(3)
        watchdog = threading.Timer(float(os.environ.get("ISK_BENCH_WATCHDOG_S", "420")), _bail)
        watchdog.daemon = True
        watchdog.start()
    if use_dp and world > 1:
        # safety valve for the first step on a real multi-GPU node: if the engine-driven RCCL run raises (on every rank alike:
        # e.g. a communicator that cannot be used), agree on it and continue on the host-driven torch.distributed hooks
        import torch
        failed = 0
        try:
            step()
        except Exception as ex:      # noqa: BLE001
            failed = 1
            print(f"[bench] rank {rank}: native data-parallel step failed ({ex}); falling back to host-driven collectives", file=sys.stderr)
        f = torch.tensor([failed], dtype=torch.int32, device=torch.device("cuda", local_rank))
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        if int(f.item()):
            local._native_agreed = False
    if args.warmup:
        step(args.warmup)
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    step(args.steps)
    sync(); barrier(); sync()
    dt = time.perf_counter() - t0
    if watchdog is not None:
        watchdog.cancel()
    if use_dp:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert os.environ.get("ISK_LIB") or all(np.isfinite(l) for l in iso.losses), "non-finite training loss"
    dp_consistent = None
    if use_dp and world > 1:
        # every rank applies the same all-reduced gradient: the reduced loss and the replicas must agree exactly
        import torch
        chk = torch.tensor([iso.losses[-1], float(np.sum(np.abs(iso.model.get_params()).astype(np.float64)))], dtype=torch.float64,
                           device=torch.device("cuda", local_rank))
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if rank == 0:
            same = bool(torch.all(lo == hi).item())
            print(f"[bench] replicas consistent across {world} ranks: {same} (loss {float(lo[0]):.6g}, |theta|_1 {float(lo[1]):.6g})", file=sys.stderr)
            dp_consistent = same

    # roofline of the dominant kernel (fused MLP forward), HIP events on the engine's stream, outside the timed region
    roof = None
    if rank == 0:
        iso.timers(enable=True, reset=True)
        step(args.profile_steps)
        sync()
        tm = iso.timers(enable=False)
        ffwd, fwd_flops, step_flops = algorithmic_flops(widths, N, K, nb_rows)
        ms, cnt = tm["fwd_ys"]
        traffic = None
        try:        # HBM bytes per launch of this kernel from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
            pm = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_summary.json")))
            if args.workload == "c3" and not args.cached_features:
                traffic = pm["mlp_fwd_traffic_bytes_per_launch"][{0: "f32", 1: "bf16x3", 2: "bf16", 3: "f16x2"}[args.forward_mode]]
        except Exception:
            traffic = None
        if cnt:
            avg = ms / cnt
            alg_tf = fwd_flops / (avg * 1e-3) / 1e12
            nt1 = (widths[1] + 15) // 16
            rows = N * K + (N if widths[-1] > 1 else 0)
            fp = (widths[0] + 31) // 32 * 32
            if args.forward_mode == 3:
                # split-f16 forward: three f16 MFMAs per product block, operands pre-split -- the kernel streams 4*Fp bytes per
                # row (f16 hi | lo planes) and is bound by that HBM stream; the executed matrix work is reported beside it
                abytes = rows * (fp * 4 + 4 + widths[-1] * 4)            # plane row + row scale read, nout floats written
                l1 = 2 * fp * (nt1 * 16)
                exec_flops = rows * (3 * l1 + (ffwd - 2 * widths[0] * widths[1]))
                ach = abytes / (avg * 1e-3) / 1e9
                roof = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                        "kernel": f"mlp_fwd_bf16x3p_kernel<{nt1},8,2,false,true>",
                        "dtype": "f16x2 split (2 scaled f16 planes per f32 operand, 3 MFMAs per block, f32 accumulate)",
                        "algorithmic_bytes_per_launch": abytes,
                        "mfma": {"executed_flops_per_launch": exec_flops, "tflops": exec_flops / (avg * 1e-3) / 1e12,
                                 "frac_of_2500_f16_peak": exec_flops / (avg * 1e-3) / 1e12 / 2500.0},
                        "algorithmic": {"flops_per_launch": fwd_flops, "tflops": alg_tf, "vs_f32_mfma_peak_157.3": alg_tf / 157.3}}
            elif args.forward_mode in (1, 2):
                # split-bf16 forward: six bf16 MFMAs per product block are EXECUTED for every algorithmic f32 one;
                # the bound is the bf16 matrix pipe, so the roofline is priced in executed bf16 flops
                l1 = 2 * ((widths[0] + 31) // 32 * 32) * (nt1 * 16)
                nmma = 6 if args.forward_mode == 1 else 1
                exec_flops = (N * K + (N if widths[-1] > 1 else 0)) * (nmma * l1 + (ffwd - 2 * widths[0] * widths[1]))
                ach = exec_flops / (avg * 1e-3) / 1e12
                roof = {"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": traffic,
                        "kernel": f"mlp_fwd_bf16x3p_kernel<{nt1},8,{3 if args.forward_mode == 1 else 1},false>",
                        "dtype": "bf16x3 split (3 bf16 planes per f32 operand, f32 accumulate)" if args.forward_mode == 1 else "bf16 operands, f32 accumulate",
                        "executed_flops_per_launch": exec_flops,
                        "algorithmic": {"flops_per_launch": fwd_flops, "tflops": alg_tf, "vs_f32_mfma_peak_157.3": alg_tf / 157.3}}
            else:
                roof = {"bound": "mfma", "achieved": alg_tf, "peak": 157.3, "unit": "TFLOP/s", "frac": alg_tf / 157.3, "traffic": traffic,
                        "kernel": f"mlp_fwd_kernel<{nt1}>", "dtype": "f32 MFMA", "flops_per_launch": fwd_flops}
            # the featurizer has the largest SUMMED duration in a kernel trace, but it runs throttled on a side stream beside
            # the training kernels (off the critical path): reported here so that the trace can be read against it
            try:
                fms, fcnt = tm["featurize"]
                if fcnt and not args.cached_features and "atoms" in c:
                    fbytes = (N * K + N) * 4 * (3 * c["atoms"] + (widths[0] + 31) // 32 * 32)
                    per_it = fms / args.profile_steps
                    roof["side_stream_kernel"] = {
                        "kernel": "featurize_reg_kernel", "bound": "hbm", "algorithmic_bytes_per_iteration": fbytes,
                        "ms_per_iteration": per_it, "achieved": fbytes / (per_it * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": fbytes / (per_it * 1e-3) / 1e9 / 8000.0,
                        "note": "deliberately throttled (one workgroup per CU) beside the latency-bound training kernels; alone it "
                                "runs at 5.4 TB/s (DESIGN.md section 3)"}
            except Exception:
                pass
            roof.setdefault("algorithmic_bytes_per_launch", rows * fp * 4)
            roof.update({"avg_ms": avg, "launches": cnt,
                         "phases_ms": {k: (v[0] / v[1] if v[1] else None) for k, v in tm.items()}})
    if use_dp:
        dist.barrier()

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.workload, args)
        value = world * N * K * args.steps / dt
        line = {
            "metric": "ISOKANN train-step Koopman-samples (N*K)/sec", "value": value, "unit": "Koopman-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {0: "f32", 3: "f32 (Koopman forward GEMM as split-f16 x2 MFMA, f32 accumulate, f32-equivalent; training f32)",
                      1: "f32 (Koopman forward GEMM as split-bf16 x3 MFMA, f32 accumulate; training f32)",
                      2: "bf16 (Koopman forward GEMM with bf16 operands, f32 accumulate: REDUCED precision, not the headline; training f32)"}[args.forward_mode],
            "data": "synthetic", "dp_replicas_consistent": dp_consistent,
            "config": {"workload": c["label"], "N_per_gpu": N, "K": K, "features": widths[0], "widths": widths,
                       "minibatch_per_gpu": args.minibatch, "optimizer": args.opt + "+weightdecay",
                       "featurize": "cached" if args.cached_features else "every-iteration (fused pdist+LayerNorm kernel)",
                       "forward_mode": {0: "f32", 1: "bf16x3", 2: "bf16", 3: "f16x2"}[args.forward_mode],
                       "parallelism": f"dp{world} over N" if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu,
            "final_loss": iso.losses[-1],
        }
        if real_stdout is not None:
            os.write(real_stdout, (json.dumps(line) + "\n").encode())
        else:
            print(json.dumps(line), flush=True)
    iso.close()
    if use_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
