"""Data-parallel ISOKANN iteration over N: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" for the CPU tests).

The reference has no multi-device code at all; this is new.  The N
axis shards naturally: every start point carries its own K replicas, so the K-mean
(src/isotarget.jl:18) never crosses a device.  Per run!() iteration the exchange steps are

  1-D target   all-reduce(max) of [-min, max] of the local Koopman values   (isotarget.jl:38)
  ISA target   all-gather of the local (d, N_r) ks and chi slices; every rank then solves the
               same tiny Float64 d x d problem (myisa + fixperm, isotarget.jl:81-127); the
               loss weights 1/std(target, dims=2) (iso.jl:183) come from all-reduced moments
  training     ONE all-reduce(sum) of the flat [gradient | batch loss] buffer per optimizer
               step (villin: 246 698 floats, 0.99 MB) -- latency-bound on xGMI, so one flat
               buffer, never per-array collectives

`minibatch` keeps the reference's meaning on the GLOBAL data set: a global batch is the union of
B/P columns from each rank's own shuffle of its shard; the gradient of the global batch mean is the
all-reduced sum of the per-rank contributions (each already divided by the global batch size).
At minibatch = 0 this is identical to the reference's single full batch.

The local engine is duck-typed (HipLocal below; the gloo tests plug the CPU oracle in), so the
collective logic is covered without a GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import capi


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous N-slab of rank `rank` (N is the slowest axis of xs (D,N) and ys (D,K,N), so a
    slab is contiguous in memory); sizes differ by at most one"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Comm:
    """torch.distributed plumbing; world == 1 works without an initialised process group"""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.on else 1
        self.rank = dist.get_rank() if self.on else 0
        self.device = device if device is not None else torch.device("cpu")

    def _t(self, a: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def allreduce_np(self, a: np.ndarray, op: str = "sum") -> np.ndarray:
        if not self.on:
            return np.array(a, copy=True)
        t = self._t(a)
        self.dist.all_reduce(t, op={"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX}[op])
        return t.cpu().numpy()

    def allgather_cols(self, a: np.ndarray, counts: List[int]) -> np.ndarray:
        """a: (d, n_r) column slice of this rank -> (d, sum n_r) in rank order"""
        if not self.on:
            return np.array(a, copy=True)
        d, nmax = a.shape[0], max(counts)
        pad = np.zeros((nmax, d), dtype=a.dtype)
        pad[:a.shape[1]] = a.T
        t = self._t(pad)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(out, counts)], axis=0).T

    def allreduce_tensor_(self, t):
        if self.on:
            if t.is_cuda and self.dist.get_backend() != "nccl":
                # a CPU process group driving GPU engines (tests): stage through the host
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def barrier(self):
        if self.on:
            self.dist.barrier()


class _DevBuf:
    """exposes engine-owned device memory to torch through __cuda_array_interface__"""
    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class HipLocal:
    """the C-ABI engine of one rank (an `Iso` built on the rank's shard)"""

    def __init__(self, iso, use_torch_stream: bool = True):
        import torch
        self.iso, self.lib, self.h = iso, iso._lib, iso._h
        self.n_local = len(iso.data)
        self.nout = iso.nout
        self.kind = "isa" if iso.target.__class__.__name__ == "TransformISA" else "shiftscale"
        self.permute = getattr(iso.target, "permute", True)
        self.whitening = getattr(iso.target, "whitening", False)
        if use_torch_stream:
            capi.check(self.lib.isk_set_stream(self.h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ptr, n = C.c_void_p(), C.c_int64()
        capi.check(self.lib.isk_dp_grad_buffer(self.h, C.byref(ptr), C.byref(n)))
        self.grad_tensor = torch.as_tensor(_DevBuf(ptr.value, n.value), device=torch.device("cuda", torch.cuda.current_device()))

    def koopman_local(self):
        capi.check(self.lib.isk_dp_koopman_local(self.h))

    def minmax_local(self):
        mm = np.zeros(2, np.float32)
        capi.check(self.lib.isk_dp_minmax_local(self.h, capi.ptr_f32(mm)))
        return float(mm[0]), float(mm[1])

    def shiftscale_apply(self, lo, hi):
        capi.check(self.lib.isk_dp_shiftscale_apply(self.h, lo, hi))

    def get_ks_chi(self):
        ks = np.empty((self.nout, self.n_local), np.float32, order="F")
        chi = np.empty((self.nout, self.n_local), np.float32, order="F")
        capi.check(self.lib.isk_dp_get_ks_chi(self.h, capi.ptr_f32(ks), capi.ptr_f32(chi)))
        return ks, chi

    def isa_solve(self, ks_all, chi_all):
        ks_all, chi_all = capi.f32(ks_all), capi.f32(chi_all)
        d, n = ks_all.shape
        At = np.zeros(d * d, np.float64)
        rp = np.zeros(d, np.int32)
        capi.check(self.lib.isk_isa_solve(capi.ptr_f32(ks_all), capi.ptr_f32(chi_all), n, d, int(self.permute),
                                          int(self.whitening), capi.ptr_f64(At), capi.ptr_i32(rp)))
        return At, rp

    def isa_device(self, comm, counts) -> bool:
        """device-resident path: all-gather the local ks / chi slices with RCCL and run the Float64
        ISA kernel on the gathered arrays on every rank (no host round trip).  Needs equal shards."""
        if len(set(counts)) != 1 or (comm.on and comm.dist.get_backend() != "nccl") or self.nout > 8:
            return False                      # (more than 8 outputs: the Float64 solve runs on the host, isk_isa_solve)
        torch = comm.torch
        ks_p, chi_p, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        capi.check(self.lib.isk_dp_ks_chi_dev(self.h, C.byref(ks_p), C.byref(chi_p), C.byref(n)))
        dev = self.grad_tensor.device
        ks_t = torch.as_tensor(_DevBuf(ks_p.value, n.value), device=dev)
        chi_t = torch.as_tensor(_DevBuf(chi_p.value, n.value), device=dev)
        if comm.on:
            gks = torch.empty(n.value * comm.world, dtype=torch.float32, device=dev)
            gchi = torch.empty_like(gks)
            comm.dist.all_gather_into_tensor(gks, ks_t)
            comm.dist.all_gather_into_tensor(gchi, chi_t)
        else:
            gks, gchi = ks_t, chi_t
        self._keep = (gks, gchi)          # alive until the kernel has run
        capi.check(self.lib.isk_dp_isa_dev(self.h, C.c_void_p(gks.data_ptr()), C.c_void_p(gchi.data_ptr()),
                                           int(sum(counts)), int(sum(counts[:comm.rank]))))
        return True

    def check_target(self):
        capi.check(self.lib.isk_dp_check_target(self.h))

    def isa_apply(self, At, rowperm):
        capi.check(self.lib.isk_dp_isa_apply(self.h, capi.ptr_f64(At), capi.ptr_i32(rowperm)))

    def target_moments(self):
        s = np.zeros(self.nout, np.float64)
        q = np.zeros(self.nout, np.float64)
        capi.check(self.lib.isk_dp_target_moments(self.h, capi.ptr_f64(s), capi.ptr_f64(q)))
        return s, q

    def set_loss_weights(self, w):
        w = np.ascontiguousarray(w, np.float32)
        capi.check(self.lib.isk_dp_set_loss_weights(self.h, capi.ptr_f32(w)))

    # ---- native collectives: RCCL called by the engine on its own stream (isk_dp_run) --------------------
    def comm_init_native(self, comm) -> bool:
        """hand the engine a communicator of its own: rank 0 draws the ncclUniqueId, torch.distributed carries the 128
        bytes.  False when the process group is not RCCL-backed (gloo tests) -- the host-driven hooks remain."""
        if getattr(self, "_native", None) is not None:
            return self._native
        self._native = False
        if not comm.on:
            self._native = True                       # single rank: isk_dp_run runs without collectives
            return True
        import os
        if (comm.dist.get_backend() != "nccl" or comm.device.type != "cuda") and not os.environ.get("ISK_RCCL_LIB"):
            return False          # (ISK_RCCL_LIB: a stand-in collective library that works across the processes of one device --
                                  #  tests/mock_rccl -- with whatever process group carries the 128-byte id)
        ident = np.zeros(128, np.uint8)
        if comm.rank == 0:
            capi.check(self.lib.isk_dp_comm_id(ident.ctypes.data_as(C.POINTER(C.c_uint8))))
        t = comm.torch.from_numpy(ident).to(comm.device)
        comm.dist.broadcast(t, src=0)
        ident = np.ascontiguousarray(t.cpu().numpy())
        capi.check(self.lib.isk_dp_comm_init(self.h, ident.ctypes.data_as(C.POINTER(C.c_uint8)), comm.rank, comm.world))
        self._native = True
        return True

    def run_native(self, n: int, epochs: int, local_batch: int, nbatches: int, global_batch: int, perms: Optional[np.ndarray]) -> np.ndarray:
        """perms: (n*epochs, N_local) int32, or None = the engine draws the shuffle orders from its seeded generator
        (while the forward runs) -> per-epoch sums of the reduced batch losses"""
        if perms is not None:
            perms = np.ascontiguousarray(perms, np.int32)
        losses = np.zeros(n * epochs, np.float64)
        capi.check(self.lib.isk_dp_run(self.h, n, epochs, int(local_batch), int(nbatches), int(global_batch),
                                       capi.ptr_i32(perms) if perms is not None else None, capi.ptr_f64(losses)))
        return losses

    def epoch_begin(self, perm):
        perm = np.ascontiguousarray(perm, np.int32)
        capi.check(self.lib.isk_dp_epoch_begin(self.h, capi.ptr_i32(perm), perm.size))

    def local_grads(self, first, nidx, global_batch):
        capi.check(self.lib.isk_dp_local_grads(self.h, int(first), int(nidx), int(global_batch)))

    def apply(self):
        capi.check(self.lib.isk_dp_apply(self.h))

    def epoch_end(self) -> float:
        loss = C.c_double(0)
        capi.check(self.lib.isk_dp_epoch_end(self.h, C.byref(loss)))
        return float(loss.value)


def dp_target(local, comm: Comm, counts: List[int]) -> None:
    """isotarget(iso) across ranks; leaves target (+ loss weights) on every rank's device"""
    local.koopman_local()
    if local.kind == "shiftscale":
        lo, hi = local.minmax_local()
        g = comm.allreduce_np(np.array([-lo, hi], np.float32), "max")     # one collective for both
        glo, ghi = -float(g[0]), float(g[1])
        if not ghi > glo:                                                 # src/isotarget.jl:39 (NaN too)
            raise capi.DomainError(capi.ECONSTANT_CHI, "Could not compute the shift-scale. chi function is constant")
        local.shiftscale_apply(glo, ghi)
        return
    if hasattr(local, "isa_device") and local.isa_device(comm, counts):
        return
    ks, chi = local.get_ks_chi()
    both = comm.allgather_cols(np.concatenate([ks, chi], axis=0), counts)  # one gather for ks and chi
    d = ks.shape[0]
    At, rp = local.isa_solve(both[:d], both[d:])
    local.isa_apply(At, rp)
    s, q = local.target_moments()
    tot = comm.allreduce_np(np.concatenate([s, q, [float(local.n_local)]]), "sum")
    n = tot[-1]
    var = (tot[d:2 * d] - tot[:d] ** 2 / n) / (n - 1)                     # corrected, src/iso.jl:183
    local.set_loss_weights(1.0 / np.sqrt(var))


def batch_plan(minibatch: int, counts: List[int], world: int, n_local: int) -> Tuple[int, int, int]:
    """(local batch, batches per epoch, global batch) of train_batch! (src/iso.jl:180-181) on the sharded data"""
    n_global = int(sum(counts))
    bs = n_global if (minibatch == 0 or n_global < minibatch) else minibatch      # src/iso.jl:180
    if bs == n_global:
        return n_local, 1, n_global                     # full batch: every rank contributes its whole shard
    bl = bs // world
    if bl < 1:
        raise ValueError("global minibatch smaller than the number of ranks")
    return bl, min(c // bl for c in counts), bl * world  # partial=false: the ragged tail is dropped


def dp_train_epoch(local, comm: Comm, minibatch: int, counts: List[int], perm: np.ndarray) -> float:
    """train_batch! (src/iso.jl:179-194) across ranks.  perm: this rank's shuffle of its own shard."""
    n_global = int(sum(counts))
    bl, nb, gb = batch_plan(minibatch, counts, comm.world, local.n_local)
    local.epoch_begin(perm[:nb * bl])
    for b in range(nb):                                 # enqueue only: no host synchronisation per step
        local.local_grads(b * bl, bl, gb)
        comm.allreduce_tensor_(local.grad_tensor)
        local.apply()
    return local.epoch_end() / n_global                 # one sync; DomainError on a non-finite global loss; src/iso.jl:193


def _native_ready(local, comm: Comm) -> bool:
    """communicator for the engine's own collectives; any failure to set it up (no librccl, init error) selects the
    host-driven hooks instead -- on every rank alike, since the outcome is agreed with one all-reduce"""
    agreed = getattr(local, "_native_agreed", None)
    if agreed is not None:
        return agreed
    try:
        ok = bool(local.comm_init_native(comm))
    except Exception:                                   # noqa: BLE001 - the fallback path is always available
        ok = False
    if comm.on:
        ok = bool(comm.allreduce_np(np.array([0.0 if ok else 1.0]), "sum")[0] == 0.0)
    local._native_agreed = ok
    return ok


def shard_counts(local, comm: Comm) -> List[int]:
    """start points per rank (one small all-reduce; callers that loop over dp_run pass the result back in)"""
    return [int(c) for c in comm.allreduce_np(np.eye(comm.world, dtype=np.int64)[comm.rank] * local.n_local, "sum")]


def dp_run(local, comm: Comm, n: int, epochs: int, minibatch: int, rng: Optional[np.random.Generator],
           perms: Optional[List[np.ndarray]] = None, counts: Optional[List[int]] = None) -> List[float]:
    """run!(iso, n, epochs) (src/iso.jl:72-94) across ranks; returns the per-epoch losses.
    Shuffle orders: `perms` (injected, parity tests), else drawn from `rng`, else (rng None) by the engine itself in the
    native run -- a host-side draw sits between two iterations with the GPU idle (~0.15 ms at N_local = 10 000)."""
    if counts is None:
        counts = shard_counts(local, comm)
    import os
    if (hasattr(local, "comm_init_native") and len(set(counts)) == 1 and os.environ.get("ISK_DP_HOST", "0") != "1"
            and (minibatch == 0 or comm.world == 1 or local.n_local * comm.world >= minibatch) and n * epochs > 0
            and not (comm.world > 1 and getattr(local, "kind", "") == "isa" and getattr(local, "nout", 0) > 8)
            and _native_ready(local, comm)):
        # equal shards on an RCCL group: the whole run inside the engine, collectives on its stream
        bl, nb, gb = batch_plan(minibatch, counts, comm.world, local.n_local)
        if perms is None and rng is None:
            ps = None
        else:
            ps = np.stack([np.asarray(perms[i] if perms is not None else rng.permutation(local.n_local), np.int32) for i in range(n * epochs)])
        return [float(l) / sum(counts) for l in local.run_native(n, epochs, bl, nb, gb, ps)]
    if rng is None:
        rng = getattr(local, "_host_rng", None) or np.random.default_rng(1234 + comm.rank)
        try:
            local._host_rng = rng
        except AttributeError:
            pass
    losses, k = [], 0
    for _ in range(n):
        dp_target(local, comm, counts)
        for _ in range(epochs):
            perm = perms[k] if perms is not None else rng.permutation(local.n_local)
            k += 1
            losses.append(dp_train_epoch(local, comm, minibatch, counts, np.asarray(perm)))
            if hasattr(local, "check_target"):
                local.check_target()                    # deferred DomainError of the device-resident target
    return losses


class IsoAll:
    """ONE host process driving several GPUs (isk_dp_create_all / isk_dp_run_all): what `gpu(iso)` of the reference's single
    Julia process becomes on a multi-GPU node (src/iso.jl:256).  `devices`: HIP ordinals, one engine and one N-slab each
    (N must be a multiple of their number).  Mirrors the part of `Iso` a training script uses: run, chis, koopman, losses."""

    def __init__(self, data, devices, *, opt=None, model=None, target=None, minibatch: int = 100,
                 featurize_every_iteration: bool = False, seed: int = 0, forward_mode: int = 0, options: Optional[dict] = None):
        from . import api
        self.data, self.devices = data, [int(d) for d in devices]
        self.optspec = opt if opt is not None else api.NesterovRegularized()
        self.modelspec = model if model is not None else api.pairnet(data)
        if target is None:
            target = api.TransformShiftscale() if api.outputdim(self.modelspec) == 1 else api.TransformISA()
        self.target, self.minibatch, self.nout = target, int(minibatch), api.outputdim(self.modelspec)
        self.losses: List[float] = []
        self._lib = capi.load()
        cfg, keep = api._make_config(data, self.modelspec, self.optspec, target, minibatch, 0, featurize_every_iteration, seed, forward_mode)
        n = len(self.devices)
        self._n = n
        self._h = (C.c_void_p * n)()
        devs = np.asarray(self.devices, np.int32)
        capi.check(self._lib.isk_dp_create_all(C.byref(cfg), capi.ptr_i32(devs), n, self._h))
        del keep
        for k, v in (options or {}).items():
            for i in range(n):
                capi.check(self._lib.isk_set_option(self._h[i], k.encode(), int(v)))
        if self.modelspec.params is None:
            self.modelspec.init_glorot(np.random.default_rng(seed))
        flat = np.ascontiguousarray(self.modelspec.params, np.float32)
        capi.check(self._lib.isk_dp_set_params_all(self._h, n, capi.ptr_f32(flat), flat.size))
        xs, ys = data.coords
        d, k, nn = ys.shape
        capi.check(self._lib.isk_dp_set_data_all(self._h, n, capi.ptr_f32(xs), capi.ptr_f32(ys), d, k, nn))
        self.n_local = nn // n

    def set_option(self, key: str, value: int, rank: Optional[int] = None):
        for i in range(self._n) if rank is None else [rank]:
            capi.check(self._lib.isk_set_option(self._h[i], key.encode(), int(value)))

    def get_option(self, key: str, rank: int = 0) -> int:
        import ctypes as C
        v = C.c_int64(0)
        capi.check(self._lib.isk_get_option(self._h[rank], key.encode(), C.byref(v)))
        return int(v.value)

    def run(self, n: int = 1, epochs: int = 1, perms=None):
        """run!(iso, n, epochs); perms: optional list (one per engine) of (n*epochs, N/n_engines) LOCAL shuffle orders"""
        losses = np.zeros(max(1, n * epochs), np.float64)
        pp = None
        if perms is not None:
            keep = [np.ascontiguousarray(p, np.int32) for p in perms]
            pp = (capi._i32p * self._n)(*[capi.ptr_i32(p) for p in keep])
        capi.check(self._lib.isk_dp_run_all(self._h, self._n, n, epochs, self.minibatch, pp, capi.ptr_f64(losses)))
        self.losses.extend(float(l) for l in losses[:n * epochs])
        return self

    def _per_engine(self, fn):
        out = []
        for i in range(self._n):
            a = np.empty((self.nout, self.n_local), np.float32, order="F")
            capi.check(fn(self._h[i], capi.ptr_f32(a)))
            out.append(a)
        return np.concatenate(out, axis=1)

    def chis(self):
        return self._per_engine(self._lib.isk_chis)

    def koopman(self):
        return self._per_engine(self._lib.isk_koopman)

    def get_params(self, rank: int = 0):
        n = self.modelspec.nparams()
        flat = np.empty(n, np.float32)
        capi.check(self._lib.isk_get_params(self._h[rank], capi.ptr_f32(flat), n))
        return flat

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._lib.isk_dp_destroy_all(self._h, self._n)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
