"""Two ranks of the data-parallel driver on ONE GPU (each process owns an engine on cuda:0 and half of the start
points; collectives through a gloo group, staged via the host): the host-driven isk_dp_* hooks at P = 2 must reproduce
the single-process isk_run on the full data with the equivalent global shuffle.  (The native driver -- collectives issued by
the engine -- runs with two ranks on one device through a mock collective library: test_gpu_dp_native2.py; real RCCL rejects
duplicate GPUs and is covered at P = 1 in test_gpu_parity.py.)"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(nout):
    from oracle import isokann_oracle as O
    A, F, N, K = 14, 40, 192, 3
    xs, ys = O.synthetic_coords(A, N, K, seed=90 + nout)
    pairs = O.random_pairs(A, F, seed=91)
    wide = bool(os.environ.get("ISK_TEST_WIDE"))          # a layered (wide) network on its bf16 path: see _mk_iso
    widths = [F, 192, 64, nout] if wide else O.pairnet_widths(F, layers=3, nout=nout)
    rng = np.random.default_rng(92)
    om = O.glorot_model(widths, rng, layernorm=not wide)
    for b in om.b:
        b[...] = 0.1 * rng.normal(size=b.shape)
    return xs, ys, pairs, widths, om.get_flat().astype(np.float32)


LR = 1e-2


def _mk_iso(pkg, xs, ys, pairs, widths, theta, nout, minibatch, every_iter=False, opt="adam"):
    wide = bool(os.environ.get("ISK_TEST_WIDE"))
    model = pkg.densenet(widths, layernorm=not wide)
    model.params = theta
    tr = pkg.TransformShiftscale() if nout == 1 else pkg.TransformISA()
    popt = pkg.AdamRegularized(LR, 1e-3) if opt == "adam" else pkg.NesterovRegularized(LR, 1e-3)
    # wide: first hidden width > 160 -> layered path; forward_mode 2 + train_bf16 = the fused bf16 forward and the bf16 training GEMMs
    # (ISK_TRAIN_BF16=0: the same in-house GEMM chain in Float32 arithmetic)
    kw = dict(forward_mode=2, options={"train_bf16": int(os.environ.get("ISK_TRAIN_BF16", "1"))}) if wide else {}
    return pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=model, opt=popt,
                   target=tr, minibatch=minibatch, featurize_every_iteration=every_iter, **kw)


def _worker(rank, world, port, nout, minibatch, niter, outdir, opt):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import importlib
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as ge
        pkg = ge.load_package()
        dp = importlib.import_module("isokann_jl_amd.dp")
        xs, ys, pairs, widths, theta = _problem(nout)
        lo, hi = dp.shard_bounds(xs.shape[1], world, rank)
        torch.cuda.set_device(0)
        iso = _mk_iso(pkg, xs[:, lo:hi], ys[:, :, lo:hi], pairs, widths, theta, nout, minibatch, opt=opt)
        local = dp.HipLocal(iso)
        comm = dp.Comm(device=torch.device("cuda", 0))
        perms = [np.random.default_rng(1000 * rank + i).permutation(hi - lo) for i in range(niter)]
        losses = dp.dp_run(local, comm, niter, 1, minibatch, None, perms=perms)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), losses=np.asarray(losses), params=iso.model.get_params(),
                 perms=np.stack(perms), lo=lo, hi=hi)
        iso.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nout,minibatch,opt", [(1, 64, "adam"), (3, 64, "adam"), (3, 0, "adam"), (1, 64, "nesterov"), (3, 64, "nesterov")])
def test_two_ranks_on_one_gpu_match_single_process(gpu_pkg, nout, minibatch, opt):
    import torch.multiprocessing as mp
    from tests.dp_checks import assert_params_match
    pkg = gpu_pkg
    world, niter = 2, 3
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), nout, minibatch, niter, d, opt), nprocs=world, join=True)
        r = [np.load(os.path.join(d, f"r{k}.npz")) for k in range(world)]
    np.testing.assert_allclose(r[0]["losses"], r[1]["losses"], rtol=1e-12)        # the reduced loss is the same number everywhere
    np.testing.assert_allclose(r[0]["params"], r[1]["params"], rtol=0, atol=0)    # and so are the replicas
    xs, ys, pairs, widths, theta = _problem(nout)
    N = xs.shape[1]
    # the equivalent global shuffle: batch b = union of the ranks' b-th local slices (src/iso.jl:180-181 on the global data)
    bs = N if (minibatch == 0 or N < minibatch) else minibatch
    bl = (r[0]["hi"] - r[0]["lo"]) if bs == N else bs // world
    nb = 1 if bs == N else min(int(x["hi"] - x["lo"]) // bl for x in r)
    gperms = []
    for it in range(niter):
        used = [np.concatenate([int(x["lo"]) + x["perms"][it][b * bl:(b + 1) * bl] for x in r]) for b in range(nb)]
        head = np.concatenate(used)
        rest = np.setdiff1d(np.arange(N), head)
        gperms.append(np.concatenate([head, rest]))
    ref = _mk_iso(pkg, xs, ys, pairs, widths, theta, nout, minibatch, opt=opt)
    pkg.run(ref, niter, 1, perms=np.stack(gperms))
    np.testing.assert_allclose(r[0]["losses"], ref.losses, rtol=2e-4)
    want = ref.model.get_params()
    ref.close()
    assert_params_match(pkg, r[0]["params"], want, opt, lambda: _mk_iso(pkg, xs, ys, pairs, widths, theta, nout, minibatch, opt=opt),
                        gperms[0][:bs], LR, niter * nb)
