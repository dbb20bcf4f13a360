"""Host-side mirror of the reference's interface for the hot path, above the C ABI.

Same names, argument meaning and error behaviour as axsk/ISOKANN.jl v2.0.0
(files relative to the reference repo):

    SimulationData(xs, ys; featurizer)       src/simulation.jl:71-76, 100-114
    FeaturesAll / FeaturesAtoms / FeaturesPairs   src/utils/features.jl:22-35
    pairnet / densenet                        src/models.jl:65-69, 87-92
    AdamRegularized / NesterovRegularized     src/models.jl:12, 20
    TransformShiftscale / TransformISA        src/isotarget.jl:32, 74-77
    Iso(data; opt, model, target, minibatch)  src/iso.jl:3-43
    run(iso, n, epochs)  [run!]               src/iso.jl:72-94
    train_batch(iso, target, ...) [train_batch!]  src/iso.jl:179-194
    isotarget / koopman                       src/isotarget.jl:10-23
    chis / chicoords                          src/iso.jl:203, 211

Arrays keep Julia's shapes: xs (D, N), ys (D, K, N), outputs (nout, N).  Atom and
pair indices are 0-based here (Python), 1-based in the reference.  Everything numeric
happens in the HIP library; this file only marshals.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import capi
from .capi import DomainError, IskError  # noqa: F401  (re-exported)


# ------------------------------------------------------------------------ featurizers --
def halfinds(n: int) -> List[Tuple[int, int]]:
    """src/utils/pairdists.jl:50-56: upper-triangle indices in column-major order"""
    return [(i, j) for j in range(1, n) for i in range(j)]


@dataclass(frozen=True)
class FeaturesAll:
    """pairwise distances between all atoms (src/utils/features.jl:22-23)"""
    def pairs(self, n_atoms):
        return None                       # the engine builds the halfinds list itself

    def featuredim(self, d):
        a = d // 3
        return a * (a - 1) // 2


@dataclass(frozen=True)
class FeaturesAtoms:
    """pairwise distances between the given atoms (src/utils/features.jl:26-29)"""
    atominds: Tuple[int, ...]

    def __init__(self, atominds):
        object.__setattr__(self, "atominds", tuple(int(i) for i in atominds))

    def pairs(self, n_atoms):
        ix = self.atominds
        return [(ix[i], ix[j]) for (i, j) in halfinds(len(ix))]

    def featuredim(self, d):
        n = len(self.atominds)
        return n * (n - 1) // 2


@dataclass(frozen=True)
class FeaturesPairs:
    """distances for an explicit pair list (src/utils/features.jl:31-34)"""
    pairlist: Tuple[Tuple[int, int], ...]

    def __init__(self, pairs):
        object.__setattr__(self, "pairlist", tuple((int(a), int(b)) for a, b in pairs))

    def pairs(self, n_atoms):
        return list(self.pairlist)

    def featuredim(self, d):
        return len(self.pairlist)


identity = None      # featurizer(::IsoSimulation) = identity  (src/simulation.jl:15)


def FeaturesCoords():
    """the coordinates themselves are the features (src/utils/features.jl:18-19: `(f::FeaturesCoords)(coords) = coords`)"""
    return identity


# ----------------------------------------------------------------------------- model ---
def pairnet_widths(n: int, layers: int = 3, nout: int = 1) -> List[int]:
    """src/models.jl:66-67 (Julia's round(Int, x) and Python's round both round half to even)"""
    return [int(round(n ** (l / layers))) for l in range(layers, 0, -1)] + [nout]


@dataclass
class ModelSpec:
    """what densenet(...) builds: Chain([LayerNorm(n)], Dense..., Dense) (src/models.jl:87-92)"""
    widths: List[int]
    activation: str = "sigmoid"
    lastactivation: str = "identity"
    layernorm: bool = False
    params: Optional[np.ndarray] = None          # flat, Flux trainables order
    ln_eps: float = 1e-5
    ln_mode: str = "var_eps2"

    def nparams(self) -> int:
        n = 2 * self.widths[0] if self.layernorm else 0
        for a, b in zip(self.widths[:-1], self.widths[1:]):
            n += a * b + b
        return n

    def init_glorot(self, rng: np.random.Generator) -> np.ndarray:
        """Flux defaults: glorot_uniform weights, zero bias, LayerNorm scale 1 / bias 0"""
        parts = []
        if self.layernorm:
            parts += [np.ones(self.widths[0], np.float32), np.zeros(self.widths[0], np.float32)]
        for fin, fout in zip(self.widths[:-1], self.widths[1:]):
            lim = math.sqrt(6.0 / (fin + fout))
            parts += [rng.uniform(-lim, lim, size=fin * fout).astype(np.float32), np.zeros(fout, np.float32)]
        self.params = np.concatenate(parts)
        return self.params


def densenet(layers: Sequence[int], activation="sigmoid", lastactivation="identity", layernorm=False,
             seed: Optional[int] = 0) -> ModelSpec:
    """src/models.jl:87-92"""
    m = ModelSpec([int(w) for w in layers], activation, lastactivation, bool(layernorm))
    m.init_glorot(np.random.default_rng(seed))
    return m


def pairnet(data=None, *, n: Optional[int] = None, layers=3, activation="sigmoid", lastactivation="identity",
            nout=1, layernorm=True, seed: Optional[int] = 0) -> ModelSpec:
    """src/models.jl:59, 65-69"""
    if n is None:
        n = featuredim(data)
    return densenet(pairnet_widths(n, layers, nout), activation, lastactivation, layernorm, seed)


def inputdim(m: ModelSpec) -> int:
    """src/models.jl:26-27"""
    return m.widths[0]


def outputdim(m: ModelSpec) -> int:
    """src/models.jl:30-31"""
    return m.widths[-1]


# ------------------------------------------------------------------------- optimisers --
@dataclass
class OptSpec:
    kind: str
    lr: float = 1e-3
    reg: float = 1e-4
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    rho: float = 0.9


def AdamRegularized(adam=1e-3, reg=1e-4) -> OptSpec:
    """src/models.jl:12: OptimiserChain(WeightDecay(reg), Adam(adam))"""
    return OptSpec("adam", adam, reg)


def NesterovRegularized(lr=1e-3, reg=1e-4) -> OptSpec:
    """src/models.jl:20: OptimiserChain(WeightDecay(reg), Nesterov(lr)); the default (src/iso.jl:18)"""
    return OptSpec("nesterov", lr, reg)


# ---------------------------------------------------------------------------- targets --
@dataclass
class TransformShiftscale:
    """src/isotarget.jl:32"""


@dataclass
class TransformISA:
    """src/isotarget.jl:74-77"""
    permute: bool = True
    whitening: bool = False


@dataclass
class TransformPseudoInv:
    """src/isotarget.jl:145-150 (experimental transform, exercised by test/runtests.jl:59).
    The d x d steps (pinv, Schur vectors, inverse) run on the host in the reference too (`|> cpu`); here chi and
    K chi come from the HIP forward (isk_chis / isk_koopman) and the target goes back through isk_train_batch."""
    normalize: bool = True
    direct: bool = True
    eigenvecs: bool = True
    permute: bool = True


def _fixperm_rows(new: np.ndarray, old: np.ndarray) -> np.ndarray:
    """fixperm (src/isotarget.jl:120-127): the row permutation of `new` closest to `old` in the entrywise 1-norm;
    ties go to the lexicographically first permutation (Combinatorics.permutations order, argmin)."""
    import itertools
    d = new.shape[0]
    best, bestp = None, None
    for p in itertools.permutations(range(d)):
        c = np.abs(new[list(p), :] - old).sum()          # norm(new[p, :] - old, 1), evaluated as the reference does:
        if best is None or c < best:                       # well separated rows make permutations tie to rounding
            best, bestp = c, p
    return new[list(bestp), :]


def _pseudoinv_target(t: TransformPseudoInv, chi: np.ndarray, kchi: np.ndarray) -> np.ndarray:
    """isotarget(::TransformPseudoInv, ...) (src/isotarget.jl:152-179) from chi = model(xs), kchi = E[model(ys)]"""
    import scipy.linalg                                                     # LAPACK gees, as Julia's schur
    chi64, k64 = chi.astype(np.float64), kchi.astype(np.float64)
    d, n = chi64.shape
    if d < 2:
        raise ValueError("TransformPseudoInv does not work with one dimensional chi functions")
    try:
        kinv = np.linalg.pinv(k64)
    except np.linalg.LinAlgError as err:
        raise capi.DomainError(capi.ESINGULAR, "Could not compute the pseudoinverse. The subspace might be singular/collapsed") from err
    if t.direct:
        m = chi64 @ kinv
        z = scipy.linalg.schur(m, output="real")[1] if t.eigenvecs else np.eye(d)
        target = z @ m @ k64
    else:
        m = k64 @ kinv
        z = scipy.linalg.schur(m, output="real")[1] if t.eigenvecs else np.eye(d)
        target = z @ np.linalg.inv(m) @ k64
    if t.normalize:
        target = target / np.abs(target).sum(axis=1, keepdims=True) * n
    if t.permute:
        target = _fixperm_rows(target, chi64)
    return np.asfortranarray(target.astype(np.float32))


# ------------------------------------------------------------------------------- data --
class SimulationData:
    """SimulationData(xs, ys; featurizer) (src/simulation.jl:100-114): coords = (xs, ys).
    The Float32 features the reference caches at construction live on the GPU, inside the
    engine that `Iso` creates; `features(data)` recomputes them through the same kernel."""

    def __init__(self, xs, ys, featurizer=identity):
        xs = np.asarray(xs)
        ys = np.asarray(ys)
        if xs.ndim != 2 or ys.ndim != 3:
            raise ValueError("xs must be (D, N) and ys (D, K, N)")
        if xs.shape[0] != ys.shape[0] or xs.shape[1] != ys.shape[2]:
            raise ValueError("xs (D, N) and ys (D, K, N) disagree")
        if featurizer is not identity and xs.shape[0] % 3 != 0:
            raise ValueError("pairwise-distance featurizers need 3*n_atoms coordinate rows")
        self.coords = (capi.f32(xs), capi.f32(ys))
        self.featurizer = featurizer

    def __len__(self):                               # src/simulation.jl:131
        return self.coords[0].shape[1]

    def __getitem__(self, i):                        # src/simulation.jl:135
        xs, ys = self.coords
        i = np.arange(len(self))[i]
        return SimulationData(xs[:, i], ys[:, :, i], self.featurizer)

    @property
    def nk(self):                                    # src/simulation.jl:129
        return self.coords[1].shape[1]


def coords(d):
    return (d.data if isinstance(d, Iso) else d).coords[0]          # src/simulation.jl:140


def propcoords(d):
    return (d.data if isinstance(d, Iso) else d).coords[1]          # src/simulation.jl:142


def featuredim(d) -> int:
    """src/simulation.jl:128"""
    d = d.data if isinstance(d, Iso) else d
    dim = d.coords[0].shape[0]
    return dim if d.featurizer is identity else d.featurizer.featuredim(dim)


# -------------------------------------------------------------------------------- Iso --
def _make_config(data: SimulationData, model: ModelSpec, opt: OptSpec, target, minibatch: int, device: int,
                 featurize_every_iteration: bool, seed: int, forward_mode: int = 0, force_layered: bool = False):
    lib = capi.load()
    cfg = capi.Config()
    lib.isk_config_default(C.byref(cfg))
    cfg.device = device
    cfg.n_layers = len(model.widths) - 1
    if cfg.n_layers > capi.ISK_MAX_LAYERS:
        raise ValueError("too many layers")
    for i, w in enumerate(model.widths):
        cfg.widths[i] = w
    cfg.activation = capi.ACT[model.activation]
    cfg.lastactivation = capi.ACT[model.lastactivation]
    cfg.layernorm = int(model.layernorm)
    cfg.ln_mode = capi.LN_MODE[model.ln_mode]
    cfg.ln_eps = model.ln_eps
    cfg.opt_kind = capi.OPT[opt.kind]
    cfg.lr, cfg.reg, cfg.beta1, cfg.beta2, cfg.opt_eps, cfg.rho = opt.lr, opt.reg, opt.beta1, opt.beta2, opt.eps, opt.rho
    keep = None
    dim = data.coords[0].shape[0]
    f = data.featurizer
    if f is identity:
        cfg.featurizer = capi.FEAT_IDENTITY
    else:
        cfg.n_atoms = dim // 3
        pl = f.pairs(cfg.n_atoms)
        if pl is None:
            cfg.featurizer = capi.FEAT_ALLPAIRS
        else:
            cfg.featurizer = capi.FEAT_PAIRLIST
            keep = np.ascontiguousarray(np.asarray(pl, dtype=np.int32).reshape(-1))
            cfg.n_pairs = len(pl)
            cfg.pairs = capi.ptr_i32(keep)
    cfg.featurize_every_iteration = int(featurize_every_iteration)
    if isinstance(target, TransformISA):
        cfg.target_kind = capi.TARGET_ISA
        cfg.isa_permute = int(target.permute)
        cfg.isa_whitening = int(target.whitening)
    elif isinstance(target, TransformShiftscale):
        cfg.target_kind = capi.TARGET_SHIFTSCALE
    elif isinstance(target, TransformPseudoInv):
        if outputdim(model) < 2:
            raise ValueError("TransformPseudoInv does not work with one dimensional chi functions")
        cfg.target_kind = capi.TARGET_HOST
    else:
        raise TypeError(f"unsupported target transform {target!r}")
    cfg.minibatch = int(minibatch)
    cfg.seed = int(seed)
    cfg.forward_mode = int(forward_mode)
    cfg.force_layered = int(bool(force_layered))
    return cfg, keep


class HipModel:
    """stands where `iso.model` (a Flux.Chain) stands: callable on (F, N) and (F, K, N)
    Float32 feature arrays (callers: src/isotarget.jl:18,101; src/iso.jl:185,203)."""

    def __init__(self, iso: "Iso"):
        self._iso = iso

    @property
    def layers(self):
        return self._iso.modelspec.widths

    def __call__(self, x):
        x = capi.f32(x)
        f, s = x.shape[0], x.shape[1:]
        ncols = int(np.prod(s)) if s else 1
        out = np.empty((self._iso.nout, ncols), dtype=np.float32, order="F")
        capi.check(self._iso._lib.isk_forward(self._iso._h, capi.ptr_f32(x), f, ncols, capi.ptr_f32(out)))
        return out.reshape((self._iso.nout,) + s, order="F")

    def get_params(self) -> np.ndarray:
        n = self._iso.modelspec.nparams()
        flat = np.empty(n, dtype=np.float32)
        capi.check(self._iso._lib.isk_get_params(self._iso._h, capi.ptr_f32(flat), n))
        return flat

    def set_params(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        capi.check(self._iso._lib.isk_set_params(self._iso._h, capi.ptr_f32(flat), flat.size))


class Iso:
    """Iso(data; opt=NesterovRegularized(), model=defaultmodel(data), target, minibatch=100)
    (src/iso.jl:3-43).  `gpu` is implied: the engine lives on `device`."""

    def __init__(self, data: SimulationData, *, opt: Optional[OptSpec] = None, model: Optional[ModelSpec] = None,
                 target=None, minibatch: int = 100, loggers=None, device: int = 0,
                 featurize_every_iteration: bool = False, seed: int = 0, forward_mode: int = 0, force_layered: bool = False,
                 options: Optional[dict] = None):
        self.data = data
        self.optspec = opt if opt is not None else NesterovRegularized()          # src/iso.jl:18
        self.modelspec = model if model is not None else pairnet(data)            # src/simulation.jl:126-127
        if self.modelspec.widths[0] != featuredim(data):
            raise ValueError("model input dimension does not match the feature dimension")
        if target is None:                                                        # src/iso.jl:28-34
            target = TransformShiftscale() if outputdim(self.modelspec) == 1 else TransformISA()
        self.target = target
        self._minibatch = int(minibatch)
        self.seed, self.forward_mode, self.featurize_every_iteration = int(seed), int(forward_mode), bool(featurize_every_iteration)
        self.koopman_weights = None
        self.losses: List[float] = []
        self.loggers = list(loggers or [])
        self.nout = outputdim(self.modelspec)
        self._lib = capi.load()
        cfg, keep = _make_config(data, self.modelspec, self.optspec, target, minibatch, device,
                                 featurize_every_iteration, seed, forward_mode, force_layered)
        h = C.c_void_p()
        capi.check(self._lib.isk_create(C.byref(cfg), C.byref(h)))
        self._h = h
        del keep
        # the engine may run another forward arithmetic than the one asked for (split-f16 falls back to split-bf16 on shapes
        # it does not cover; ISK_FWD_MODE overrides at creation): keep what it reports, so that save() records what ran
        self.forward_mode = self.get_option("forward_mode")
        for k, v in (options or {}).items():
            self.set_option(k, v)
        self.model = HipModel(self)
        if self.modelspec.params is None:
            self.modelspec.init_glorot(np.random.default_rng(seed))
        self.model.set_params(self.modelspec.params)
        xs, ys = data.coords
        d, k, n = ys.shape
        capi.check(self._lib.isk_set_data(self._h, capi.ptr_f32(xs), capi.ptr_f32(ys), d, k, n))

    @property
    def minibatch(self) -> int:
        """Iso.minibatch is a mutable field in the reference (src/iso.jl:10): assigning it re-plans the engine's run!()"""
        return self._minibatch

    @minibatch.setter
    def minibatch(self, mb: int):
        if getattr(self, "_h", None):
            capi.check(self._lib.isk_set_minibatch(self._h, int(mb)))      # validates: the field follows only on success
        self._minibatch = int(mb)

    # -- engine options (kernel variants / scheduling; include/isokann_hip.h lists the keys) ----------------------------
    def set_option(self, key: str, value: int) -> None:
        capi.check(self._lib.isk_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = C.c_int64(0)
        capi.check(self._lib.isk_get_option(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.isk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- optimizer state (Flux.setup / save / load) -------------------------------------
    def get_opt_state(self):
        n = self.modelspec.nparams()
        m = np.empty(n, np.float32)
        v = np.empty(n, np.float32)
        bt = np.zeros(2, np.float64)
        capi.check(self._lib.isk_get_opt_state(self._h, capi.ptr_f32(m), capi.ptr_f32(v), n, capi.ptr_f64(bt)))
        return m, v, bt

    def set_opt_state(self, m, v, bt):
        m = np.ascontiguousarray(m, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        bt = np.ascontiguousarray(bt, np.float64)
        capi.check(self._lib.isk_set_opt_state(self._h, capi.ptr_f32(m), capi.ptr_f32(v), m.size, capi.ptr_f64(bt)))

    # -- instrumentation ----------------------------------------------------------------
    def timers(self, enable: Optional[bool] = None, reset: bool = False):
        if enable is not None:
            capi.check(self._lib.isk_timers_enable(self._h, int(enable)))
        ms = np.zeros(capi.T_COUNT, np.float64)
        cnt = np.zeros(capi.T_COUNT, np.int64)
        capi.check(self._lib.isk_timers_read(self._h, capi.ptr_f64(ms), cnt.ctypes.data_as(C.POINTER(C.c_int64))))
        if reset:
            capi.check(self._lib.isk_timers_reset(self._h))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(capi.TIMER_NAMES)}


def _perm_array(perm, n):
    if perm is None:
        return None
    p = np.ascontiguousarray(perm, dtype=np.int32)
    if p.shape[-1] != n:
        raise ValueError("permutation length must equal the number of samples")
    return p


def run(iso: Iso, n: int = 1, epochs: int = 1, perms=None, showprogress: bool = False) -> Iso:
    """run!(iso, n, epochs) (src/iso.jl:72-94).  `perms`: optional (n*epochs, N) injected
    DataLoader shuffle orders (parity tests); otherwise the engine's seeded shuffle."""
    N = len(iso.data)
    p = _perm_array(perms, N)
    if p is not None and p.size != n * epochs * N:
        raise ValueError("need one permutation per (iteration, epoch)")
    if isinstance(iso.target, TransformPseudoInv):
        # host-side transform: forward passes and training on the GPU, the d x d algebra here (src/iso.jl:72-94)
        pr = None if p is None else p.reshape(n, epochs, N)
        for it in range(n):
            target = isotarget(iso)
            for ep in range(epochs):
                iso.losses.append(train_batch(iso, target, perm=None if pr is None else pr[it, ep]))
            for logger in iso.loggers:
                logger(iso=iso)
        return iso
    if not iso.loggers:
        losses = np.zeros(n * epochs, np.float64)
        capi.check(iso._lib.isk_run(iso._h, n, epochs, capi.ptr_i32(p) if p is not None else None, capi.ptr_f64(losses)))
        iso.losses.extend(float(l) for l in losses)
        return iso
    # loggers run after every outer iteration on the host (src/iso.jl:84-91)
    p2 = None if p is None else p.reshape(n, epochs, N)
    for it in range(n):
        losses = np.zeros(epochs, np.float64)
        pi = None if p2 is None else np.ascontiguousarray(p2[it])
        capi.check(iso._lib.isk_run(iso._h, 1, epochs, capi.ptr_i32(pi) if pi is not None else None, capi.ptr_f64(losses)))
        iso.losses.extend(float(l) for l in losses)
        for logger in iso.loggers:
            logger(iso=iso)
    return iso


def isotarget(iso: Iso) -> np.ndarray:
    """isotarget(iso) (src/isotarget.jl:10-12) -> (nout, N)"""
    if isinstance(iso.target, TransformPseudoInv):
        return _pseudoinv_target(iso.target, chis(iso), koopman(iso))
    out = np.empty((iso.nout, len(iso.data)), np.float32, order="F")
    capi.check(iso._lib.isk_target(iso._h, capi.ptr_f32(out)))
    return out


def train_batch(iso: Iso, target=None, minibatch: Optional[int] = None, perm=None) -> float:
    """train_batch!(model, xs, target, opt, minibatch) (src/iso.jl:179-194) on the resident xs.
    target=None reuses the device-resident target of the last isotarget(iso)."""
    t = None if target is None else capi.f32(target)
    if t is not None and t.shape != (iso.nout, len(iso.data)):
        raise ValueError("target must be (nout, N)")
    p = _perm_array(perm, len(iso.data))
    loss = C.c_double(0)
    capi.check(iso._lib.isk_train_batch(iso._h, capi.ptr_f32(t) if t is not None else None,
                                        iso.minibatch if minibatch is None else minibatch,
                                        capi.ptr_i32(p) if p is not None else None, C.byref(loss)))
    return float(loss.value)


def batch_grads(iso: Iso, target, idx) -> Tuple[np.ndarray, float]:
    """gradient of l/numobs for one batch (what Zygote hands to update!, src/iso.jl:184-191),
    flat in Flux trainables order, and the batch loss l"""
    t = None if target is None else capi.f32(target)
    idx = np.ascontiguousarray(idx, np.int32)
    g = np.empty(iso.modelspec.nparams(), np.float32)
    loss = C.c_double(0)
    capi.check(iso._lib.isk_batch_grads(iso._h, capi.ptr_f32(t) if t is not None else None, capi.ptr_i32(idx), idx.size,
                                        capi.ptr_f32(g), C.byref(loss)))
    return g, float(loss.value)


def chis(iso: Iso) -> np.ndarray:
    """chis(iso) (src/iso.jl:203) -> (nout, N)"""
    out = np.empty((iso.nout, len(iso.data)), np.float32, order="F")
    capi.check(iso._lib.isk_chis(iso._h, capi.ptr_f32(out)))
    return out


def koopman(iso: Iso) -> np.ndarray:
    """koopman(iso) = expectation(model, propfeatures) (src/isotarget.jl:18,20) -> (nout, N)"""
    out = np.empty((iso.nout, len(iso.data)), np.float32, order="F")
    capi.check(iso._lib.isk_koopman(iso._h, capi.ptr_f32(out)))
    return out


def koopman_f64(iso: Iso) -> np.ndarray:
    """K chi before its rounding to Float32 (the K-means of the Float32 outputs summed in Float64): what the engine's TransformISA
    solve reads (isk_koopman_f64)"""
    out = np.empty((iso.nout, len(iso.data)), np.float64, order="F")
    capi.check(iso._lib.isk_koopman_f64(iso._h, capi.ptr_f64(out)))
    return out


def chicoords(iso: Iso, xs) -> np.ndarray:
    """chicoords(iso, xs) (src/iso.jl:211): featurize raw coordinates, then the model"""
    x = capi.f32(xs)
    d, s = x.shape[0], x.shape[1:]
    ncols = int(np.prod(s)) if s else 1
    out = np.empty((iso.nout, ncols), np.float32, order="F")
    capi.check(iso._lib.isk_chicoords(iso._h, capi.ptr_f32(x), d, ncols, capi.ptr_f32(out)))
    return out.reshape((iso.nout,) + s, order="F")


def features(iso: Iso, coords_=None) -> np.ndarray:
    """features(data) / features(data, coords) (src/simulation.jl:121-124, 141): the featurizer
    applied to coords (default: the resident xs)"""
    x = capi.f32(iso.data.coords[0] if coords_ is None else coords_)
    d, s = x.shape[0], x.shape[1:]
    ncols = int(np.prod(s)) if s else 1
    F = featuredim(iso)
    out = np.empty((F, ncols), np.float32, order="F")
    capi.check(iso._lib.isk_features(iso._h, capi.ptr_f32(x), d, ncols, capi.ptr_f32(out)))
    return out.reshape((F,) + s, order="F")


def propfeatures(iso: Iso) -> np.ndarray:
    """propfeatures(data) (src/simulation.jl:143)"""
    return features(iso, iso.data.coords[1])


# --------------------------------------------------- evaluation callers --
def koopman_coords(iso: Iso, ys) -> np.ndarray:
    """expectation(model, featurizer(ys)) (src/isotarget.jl:18) for coordinates ys (D, K, n) -> (nout, n)"""
    y = capi.f32(ys)
    d, k, n = y.shape
    out = np.empty((iso.nout, n), np.float32, order="F")
    capi.check(iso._lib.isk_koopman_coords(iso._h, capi.ptr_f32(y), d, k, n, capi.ptr_f32(out)))
    return out


def chi_kchi(iso: Iso):
    """chi_kchi(iso) (src/isotarget.jl:22-23)"""
    return chis(iso), koopman(iso)


def validationloss(iso: Iso, valdata: SimulationData) -> float:
    """validationloss(iso, valdata) (src/iso.jl:160-168)"""
    vx, vy = valdata.coords
    d, k, n = vy.shape
    loss = C.c_double(0)
    capi.check(iso._lib.isk_validationloss(iso._h, capi.ptr_f32(vx), capi.ptr_f32(vy), d, k, n, C.byref(loss)))
    return float(loss.value)


class ValidationLossLogger:
    """ValidationLossLogger(data; logevery=10) (src/iso.jl:132-177): a logger for Iso(...; loggers=[...])"""

    def __init__(self, data: SimulationData, logevery: int = 10):
        self.data, self.logevery = data, logevery
        self.losses: List[float] = []
        self.iters: List[int] = []

    def __call__(self, iso: Iso, **kw):                                   # log!(v; iso)
        if len(iso.losses) % self.logevery != 0:
            return
        self.losses.append(validationloss(iso, self.data))
        self.iters.append(len(iso.losses))

    def diagnostic(self):
        return ("validation loss", self.losses[-1] if self.losses else None)


def koopman_variance(iso: Iso, ys=None) -> float:
    """koopman_variance(iso, ys) (src/iso.jl:355-359)"""
    chi = chicoords(iso, iso.data.coords[1] if ys is None else ys)
    i, k, n = chi.shape
    return float(np.sum((chi - chi.sum(axis=1, keepdims=True) / k) ** 2) / i / n)


def rates(iso_or_x, y=None, lagtime: float = 1.0) -> np.ndarray:
    """rates(iso) / rates(x, y) (src/iso.jl:339-351): the coarse-grained rate matrix Q with K chi = exp(tau Q) chi,
    log(y / x) (Julia's right division: least squares) divided by the lag time; the 1-D case is extended by 1 - chi.
    chi and K chi come from the HIP forward; the d x d matrix logarithm runs on the host as in the reference (`cpu`)."""
    import scipy.linalg
    if y is None:
        x, y = chis(iso_or_x), koopman(iso_or_x)
    else:
        x = np.asarray(iso_or_x)
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    if x.shape[0] == 1:
        x = np.concatenate([x, 1 - x], axis=0)
        y = np.concatenate([y, 1 - y], axis=0)
    q = np.linalg.lstsq(x.T, y.T, rcond=None)[0].T                      # y / x
    return scipy.linalg.logm(q) / lagtime


# ------------------------------------- checkpoint / data growth --
def lastcat(a, b):
    """lastcat(x, y) (src/data.jl:59-62): concatenate along the last (sample) axis"""
    return np.concatenate([np.asarray(a), np.asarray(b)], axis=-1)


def mergedata(d1: SimulationData, d2: SimulationData) -> SimulationData:
    """mergedata(d1, d2) (src/simulation.jl:162-171): coords of both, featurizer of d1"""
    return SimulationData(lastcat(d1.coords[0], d2.coords[0]), lastcat(d1.coords[1], d2.coords[1]), d1.featurizer)


def _data_delta(old: SimulationData, new: SimulationData):
    """(drop_first, n_add) when `new` is `old` minus its first start points plus appended ones -- what addcoords! and the cutoff
    window of run_kde! produce (src/iso.jl:238,288-290) -- else None"""
    (oxs, oys), (nxs, nys) = old.coords, new.coords
    if oys.shape[:2] != nys.shape[:2] or oxs.shape[0] != nxs.shape[0]:
        return None
    no, nn = oxs.shape[1], nxs.shape[1]
    if nn >= no and np.array_equal(nxs[:, :no], oxs) and np.array_equal(nys[:, :, :no], oys):
        return 0, nn - no                                   # mergedata(old, extra)
    if nn < no and np.array_equal(nxs, oxs[:, no - nn:]) and np.array_equal(nys, oys[:, :, no - nn:]):
        return no - nn, 0                                   # old[end-cutoff+1:end]
    return None


def append_data(iso: Iso, xs_new=None, ys_new=None, drop_first: int = 0) -> Iso:
    """addcoords! / the cutoff window (src/iso.jl:238,288-290) without re-uploading the resident start points: drop the first
    `drop_first`, append the columns of xs_new (D, n), ys_new (D, K, n).  Only the new columns cross PCIe (isk_append_data)."""
    oxs, oys = iso.data.coords
    d, k, _ = oys.shape
    n_add = 0 if xs_new is None else int(np.asarray(xs_new).shape[1])
    if n_add:
        xs_new = np.asfortranarray(np.asarray(xs_new, np.float32))
        ys_new = np.asfortranarray(np.asarray(ys_new, np.float32))
        if xs_new.shape[0] != d or ys_new.shape[:2] != (d, k) or ys_new.shape[2] != n_add:
            raise ValueError("new samples must be xs (D, n), ys (D, K, n) with the D and K of the resident data")
    capi.check(iso._lib.isk_append_data(iso._h, capi.ptr_f32(xs_new) if n_add else None, capi.ptr_f32(ys_new) if n_add else None,
                                        d, k, n_add, int(drop_first)))
    keep_x, keep_y = oxs[:, drop_first:], oys[:, :, drop_first:]
    iso.data = SimulationData(lastcat(keep_x, xs_new) if n_add else keep_x, lastcat(keep_y, ys_new) if n_add else keep_y, iso.data.featurizer)
    return iso


def set_data(iso: Iso, data: SimulationData) -> Iso:
    """`iso.data = newdata` (src/iso.jl:238,289): model and optimizer state stay.  Data that only grew at the end or lost start
    points at the front (addcoords!, the cutoff window) keep their resident part on the device (isk_append_data); anything else is
    re-uploaded and re-featurized"""
    if featuredim(data) != iso.modelspec.widths[0]:
        raise ValueError("feature dimension of the new data does not match the model")
    delta = _data_delta(iso.data, data) if iso.data is not None else None
    if delta is not None:
        drop, n_add = delta
        no = iso.data.coords[0].shape[1]
        append_data(iso, data.coords[0][:, no - drop:] if n_add else None, data.coords[1][:, :, no - drop:] if n_add else None, drop)
        iso.data = data
        return iso
    xs, ys = data.coords
    d, k, n = ys.shape
    capi.check(iso._lib.isk_set_data(iso._h, capi.ptr_f32(xs), capi.ptr_f32(ys), d, k, n))
    iso.data = data
    return iso


def _featurizer_state(f):
    if f is identity:
        return ("identity", None)
    if isinstance(f, FeaturesAll):
        return ("all", None)
    if isinstance(f, FeaturesAtoms):
        return ("atoms", np.asarray(f.atominds, np.int64))
    if isinstance(f, FeaturesPairs):
        return ("pairs", np.asarray(f.pairlist, np.int64))
    raise TypeError(f)


def save(path: str, iso: Iso) -> None:
    """save(path, iso) (src/iso.jl:405-408): cpu(iso) then one file with model, optimizer state, data,
    target and losses.  Parameters and Adam/Nesterov state are stored flat in Flux trainables order
    (W column-major), the layout a Julia host needs to rebuild a Flux.Chain + Optimisers state tree."""
    m, v, bt = iso.get_opt_state()
    fk, fv = _featurizer_state(iso.data.featurizer)
    t = iso.target
    np.savez(path, format_version=1,
             widths=np.asarray(iso.modelspec.widths), activation=iso.modelspec.activation,
             lastactivation=iso.modelspec.lastactivation, layernorm=iso.modelspec.layernorm,
             ln_eps=iso.modelspec.ln_eps, ln_mode=iso.modelspec.ln_mode, params=iso.model.get_params(),
             opt_kind=iso.optspec.kind, opt_hyper=np.asarray([iso.optspec.lr, iso.optspec.reg, iso.optspec.beta1,
                                                             iso.optspec.beta2, iso.optspec.eps, iso.optspec.rho]),
             opt_m=m, opt_v=v, opt_bt=bt,
             xs=iso.data.coords[0], ys=iso.data.coords[1], featurizer_kind=fk,
             featurizer_arg=np.zeros(0) if fv is None else fv,
             target_kind=type(t).__name__,
             target_args=np.asarray([getattr(t, "permute", True), getattr(t, "whitening", False), getattr(t, "normalize", True),
                                     getattr(t, "direct", True), getattr(t, "eigenvecs", True)]),
             minibatch=iso.minibatch, losses=np.asarray(iso.losses, np.float64),
             # engine configuration of this host (no counterpart in the reference's Iso): without it a reloaded Iso would run
             # another forward arithmetic / featurization mode / shuffle stream
             seed=iso.seed, forward_mode=iso.forward_mode, featurize_every_iteration=iso.featurize_every_iteration,
             koopman_weights=np.zeros((0, 0), np.float32) if iso.koopman_weights is None else iso.koopman_weights)


def load(path: str, **iso_kwargs) -> Iso:
    """load(path) (src/iso.jl:417-420)"""
    z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False)
    fk = str(z["featurizer_kind"])
    feat = {"identity": lambda a: identity, "all": lambda a: FeaturesAll(), "atoms": lambda a: FeaturesAtoms(a),
            "pairs": lambda a: FeaturesPairs(a)}[fk](z["featurizer_arg"])
    model = ModelSpec([int(w) for w in z["widths"]], str(z["activation"]), str(z["lastactivation"]), bool(z["layernorm"]),
                      z["params"].astype(np.float32), float(z["ln_eps"]), str(z["ln_mode"]))
    h = z["opt_hyper"]
    opt = OptSpec(str(z["opt_kind"]), *[float(x) for x in h])
    tk, ta = str(z["target_kind"]), z["target_args"]
    if tk == "TransformISA":
        target = TransformISA(bool(ta[0]), bool(ta[1]))
    elif tk == "TransformPseudoInv":
        target = TransformPseudoInv(bool(ta[2]), bool(ta[3]), bool(ta[4]), bool(ta[0]))
    else:
        target = TransformShiftscale()
    for k in ("seed", "forward_mode", "featurize_every_iteration"):          # keyword arguments override what the file holds
        if k in z.files and k not in iso_kwargs:
            iso_kwargs[k] = z[k].item()
    iso = Iso(SimulationData(z["xs"], z["ys"], feat), opt=opt, model=model, target=target, minibatch=int(z["minibatch"]), **iso_kwargs)
    iso.set_opt_state(z["opt_m"], z["opt_v"], z["opt_bt"])
    if "koopman_weights" in z.files and z["koopman_weights"].size:
        set_koopman_weights(iso, np.asarray(z["koopman_weights"]).reshape(-1, order="F"))
    iso.losses = [float(l) for l in z["losses"]]
    return iso


# ------------------------------------------------ d chi / d coords --
def chicoords_pullback(iso: Iso, xs, cot) -> np.ndarray:
    """Zygote pullback of chicoords(iso, xs) (src/iso.jl:211): (d chi / d xs)' * cot, cot (nout, n) -> (D, n).
    What `Zygote.gradient(x -> sum(cot .* chicoords(iso, x)), xs)` returns in the reference
    (src/utils/minimumpath.jl:3-13, src/simulators/metadynamics.jl:40-43)."""
    x = capi.f32(xs)
    if x.ndim == 1:
        x = x.reshape(-1, 1, order="F")
    d, n = x.shape
    c = capi.f32(np.asarray(cot).reshape(iso.nout, n, order="F"))
    out = np.empty((d, n), np.float32, order="F")
    capi.check(iso._lib.isk_chicoords_pullback(iso._h, capi.ptr_f32(x), d, n, capi.ptr_f32(c), capi.ptr_f32(out)))
    return out


def dchidx(iso: Iso, x) -> np.ndarray:
    """Jacobian d chi / d x of chicoords at ONE coordinate vector x (D,) -> (nout, D) (dchidx, src/simulators/openmm.jl:659-664)"""
    x = np.asarray(x, np.float32).reshape(-1)
    xs = np.repeat(x[:, None], iso.nout, axis=1)
    return chicoords_pullback(iso, xs, np.eye(iso.nout, dtype=np.float32)).T


# -------------------------- trajectory builders and weighted samples --
def data_from_trajectory(xs, reverse: bool = True, stride: int = 1, lag: int = 1):
    """data_from_trajectory(xs; reverse, stride, lag) (src/data.jl:88-100): (x, y) pairs from a trajectory
    xs (D, T); with reverse=true each start gets the two time-symmetric partners (K = 2)"""
    xs = np.asarray(xs)
    n = xs.shape[1]
    if reverse:
        rng = np.arange(lag, n - lag, stride)
        ys = np.stack([xs[:, rng - lag], xs[:, rng + lag]], axis=1)
    else:
        rng = np.arange(0, n - lag, stride)
        ys = xs[:, rng + lag][:, None, :]
    return xs[:, rng], ys


def data_from_trajectories(xss, **kwargs):
    """data_from_trajectories(xss; kwargs...) (src/data.jl:111-130)"""
    if isinstance(xss, np.ndarray) and xss.ndim == 3:
        xss = [xss[:, :, i] for i in range(xss.shape[2])]
    datas = [data_from_trajectory(x, **kwargs) for x in xss]
    return np.concatenate([d[0] for d in datas], axis=1), np.concatenate([d[1] for d in datas], axis=2)


@dataclass
class WeightedSamples:
    """WeightedSamples(values, weights) (src/data.jl:187-190): ys with Girsanov weights (1, K, N)"""
    values: np.ndarray
    weights: np.ndarray


def set_koopman_weights(iso: Iso, weights) -> None:
    """use the weighted expectation (src/data.jl:215) for the resident ys: sum_k w f(y_k) / K; None resets"""
    if weights is None:
        capi.check(iso._lib.isk_set_koopman_weights(iso._h, None, 0, 0))
        iso.koopman_weights = None
        return
    w = capi.f32(np.asarray(weights).reshape(iso.data.nk, len(iso.data), order="F"))
    capi.check(iso._lib.isk_set_koopman_weights(iso._h, capi.ptr_f32(w), w.shape[0], w.shape[1]))
    iso.koopman_weights = np.array(w, copy=True)
