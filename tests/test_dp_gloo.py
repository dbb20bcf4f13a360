"""World-size-2 `gloo` test of the data-parallel driver (isokann.jl_amd/dp.py) on CPU.

The collective logic (target statistics, ISA gather, one flat gradient all-reduce per
optimizer step, global-batch semantics) is exercised with the CPU oracle plugged in as the
per-rank local engine; the sharded result must equal the single-process oracle run on the
full data with the equivalent global shuffle."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleLocal:
    """dp.py local-engine protocol on top of the NumPy oracle (test-only)"""

    def __init__(self, O, model, rule, xs, ys, kind, permute=True, whitening=False):
        self.O, self.m, self.st = O, model, O.setup(rule, model)
        self.xs, self.ys = xs, ys
        self.n_local, self.nout, self.kind = xs.shape[1], model.widths[-1], kind
        self.permute, self.whitening = permute, whitening
        self.P = model.nparams()
        self.grad_tensor = torch.zeros(self.P + 1, dtype=torch.float64)
        self.w = np.ones(self.nout)

    def koopman_local(self):
        self.ks, self.chi = self.O.expectation(self.m, self.ys), self.m(self.xs)

    def minmax_local(self):
        return float(self.ks.min()), float(self.ks.max())

    def shiftscale_apply(self, lo, hi):
        # the driver moves min/max as float32 (as the GPU engine does); do the same on the reference side
        self.target = (self.ks - lo) / (hi - lo)

    def get_ks_chi(self):
        return self.ks, self.chi

    def isa_solve(self, ks_all, chi_all):
        O = self.O
        A = O.myisa(ks_all.T, self.whitening)
        t = A.T @ ks_all
        rp = O.fixperm_index(t, chi_all) if self.permute else tuple(range(t.shape[0]))
        return A.T.reshape(-1).copy(), np.asarray(rp, np.int32)

    def isa_apply(self, At, rp):
        d = self.nout
        self.target = (At.reshape(d, d) @ self.ks)[list(rp)]

    def target_moments(self):
        return self.target.sum(axis=1), (self.target ** 2).sum(axis=1)

    def set_loss_weights(self, w):
        self.w = np.asarray(w, np.float64)

    def epoch_begin(self, perm):
        self.perm, self.lsum, self.dead = np.asarray(perm), 0.0, False

    def epoch_end(self):
        if self.dead:
            raise self.O.DomainError("collapsed")
        return self.lsum

    def local_grads(self, first, nidx, gb):
        O = self.O
        idx = self.perm[first:first + nidx]
        out, cache = O.forward_cache(self.m, self.xs[:, idx])
        r = (out - self.target[:, idx]) * self.w[:, None]
        grads = O.backward(self.m, cache, 2 * r * self.w[:, None] / gb)
        flat = np.concatenate([g.reshape(-1, order="F") for g in grads])
        self.grad_tensor[:self.P] = torch.from_numpy(flat)
        self.grad_tensor[self.P] = float(np.sum(r * r))

    def apply(self):
        l = float(self.grad_tensor[self.P])
        if self.dead or not np.isfinite(l):
            self.dead = True
            return
        self.lsum += l
        flat = self.grad_tensor[:self.P].numpy()
        grads, off = [], 0
        for a in self.m.arrays():
            grads.append(flat[off:off + a.size].reshape(a.shape, order="F"))
            off += a.size
        self.O.update(self.st, self.m, grads)


def _case(nout, n=53):
    from oracle import isokann_oracle as O
    rng = np.random.default_rng(100 + nout)
    F, K = 12, 3
    xs = rng.normal(size=(F, n)) + 1
    ys = xs[:, None, :] + 0.2 * rng.normal(size=(F, K, n))
    m = O.glorot_model([F, 6, 4, nout], rng)
    return O, xs, ys, m


def _worker(rank, world, port, nout, minibatch, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as ge
        dp = ge.load_package().dp if hasattr(ge.load_package(), "dp") else None
        if dp is None:
            import importlib
            dp = importlib.import_module("isokann_jl_amd.dp")
        O, xs, ys, m = _case(nout)
        n = xs.shape[1]
        lo, hi = dp.shard_bounds(n, world, rank)
        kind = "shiftscale" if nout == 1 else "isa"
        local = OracleLocal(O, m.copy(), O.AdamRegularized(1e-2, 1e-3), xs[:, lo:hi], ys[:, :, lo:hi], kind)
        comm = dp.Comm()
        assert comm.world == world and comm.rank == rank
        rng = np.random.default_rng(7 + rank)
        perms = [rng.permutation(hi - lo) for _ in range(3)]
        losses = dp.dp_run(local, comm, 3, 1, minibatch, rng, perms=perms)
        np.savez(os.path.join(outdir, f"r{rank}.npz"), losses=np.asarray(losses), params=local.m.get_flat(),
                 perms=np.stack(perms), lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nout,minibatch", [(1, 0), (1, 16), (3, 0), (3, 20)])
def test_two_rank_gloo_equals_single_process(nout, minibatch):
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), nout, minibatch, d), nprocs=world, join=True)
        res = [np.load(os.path.join(d, f"r{r}.npz")) for r in range(world)]
    import __graft_entry__ as ge
    ge.load_package()
    import importlib
    dp = importlib.import_module("isokann_jl_amd.dp")
    O, xs, ys, m = _case(nout)
    n = xs.shape[1]
    # both ranks hold the same model and report the same global loss
    np.testing.assert_allclose(res[0]["params"], res[1]["params"], rtol=0, atol=0)
    np.testing.assert_allclose(res[0]["losses"], res[1]["losses"], rtol=0, atol=0)
    # single-process reference with the equivalent global shuffle
    counts = [int(r["hi"] - r["lo"]) for r in res]
    bs = n if (minibatch == 0 or n < minibatch) else minibatch
    ref = m.copy()
    st = O.setup(O.AdamRegularized(1e-2, 1e-3), ref)
    tr = O.TransformShiftscale() if nout == 1 else O.TransformISA()
    ref_losses = []
    for it in range(3):
        tgt = O.isotarget(tr, ref, xs, ys)
        if nout == 1:
            # the driver exchanges min/max as float32
            ks = O.expectation(ref, ys)
            lo32, hi32 = np.float32(ks.min()), np.float32(ks.max())
            tgt = (ks - float(lo32)) / (float(hi32) - float(lo32))
        if bs == n:
            gperm = np.concatenate([r["lo"] + r["perms"][it] for r in res])
            ref_losses.append(O.train_batch(ref, xs, tgt, st, 0, perm=gperm))
        else:
            bl = bs // world
            nb = min(c // bl for c in counts)
            w = O.loss_weights(tgt)
            ls = 0.0
            for b in range(nb):
                idx = np.concatenate([r["lo"] + r["perms"][it][b * bl:(b + 1) * bl] for r in res])
                l, g, _ = O.batch_loss_and_grads(ref, xs[:, idx], tgt[:, idx], w)
                ls += l
                O.update(st, ref, g)
            ref_losses.append(ls / n)
    np.testing.assert_allclose(res[0]["losses"], ref_losses, rtol=1e-9)
    np.testing.assert_allclose(res[0]["params"], ref.get_flat(), rtol=1e-8, atol=1e-11)


def test_shard_bounds_cover_and_balance():
    import __graft_entry__ as ge
    ge.load_package()
    import importlib
    dp = importlib.import_module("isokann_jl_amd.dp")
    for n in (1, 7, 8, 10000, 10001):
        for w in (1, 2, 3, 8):
            b = [dp.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
