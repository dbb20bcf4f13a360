"""CPU restatement of the ISOKANN.jl hot path (NumPy).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (axsk/ISOKANN.jl v2.0.0, pure Julia) can be neither
run nor imported in this container (no Julia toolchain) and none of its own tests
pins a number on this path (every assertion is `@test true`, test/runtests.jl:18,28,
53,62,77).  This file is therefore a restatement of the reference's algorithm read
from its source, plus the published semantics of the pinned third-party packages
it delegates to (Flux 0.16.9, Optimisers 0.4.7, NNlib 0.9.34, MLUtils 0.4.8,
StatsBase 0.34.10, PCCAPlus 1.1.2, Combinatorics 1.1.0 -- Manifest-v1.12.toml).
What pins it instead (tests/test_oracle.py): analytic known answers, an
independent PyTorch-CPU autograd cross-check of loss/gradients/optimizer steps,
and committed golden fixtures generated here (tests/golden/).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / reported CPU baseline.  The product path
(isokann.jl_amd) never imports it.

Array conventions follow Julia: xs is (D, N), ys is (D, K, N), model outputs are
(nout, N) / (nout, K, N), weights W are (out, in).  The flat parameter vector is
in Flux trainables order [LN.scale, LN.bias, W1 (column-major), b1, W2, b2, ...].

All line citations are relative to /root/reference/.
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np


class DomainError(ValueError):
    """Julia's DomainError; raised at the three places the reference raises it
    (src/isotarget.jl:39, src/isotarget.jl:94-97, src/iso.jl:186-189)."""


# ----------------------------------------------------------------------------------
# featurizers  (src/utils/pairdists.jl, src/utils/features.jl)
# ----------------------------------------------------------------------------------

def halfinds(n: int) -> List[Tuple[int, int]]:
    """0-based (i, j), i < j, in the order of src/utils/pairdists.jl:50-56:
    `findall` over an upper-triangular mask walks column-major, i.e.
    (1,2),(1,3),(2,3),(1,4),... in Julia's 1-based terms."""
    return [(i, j) for j in range(1, n) for i in range(j)]


def _cols(x: np.ndarray) -> Tuple[np.ndarray, Tuple[int, ...]]:
    """View x::(d, s...) as (d, prod(s)) in Julia (column-major) column order."""
    d, s = x.shape[0], x.shape[1:]
    return x.reshape(d, -1, order="F"), s


def pdists(coords: np.ndarray, inds: Sequence[Tuple[int, int]]) -> np.ndarray:
    """Distances for an explicit list of 0-based atom pairs.
    src/utils/pairdists.jl:109-127 (`pdists`), called by FeaturesPairs
    (src/utils/features.jl:31-34).  Direct-difference form."""
    b, s = _cols(coords)
    a_idx = np.asarray([p[0] for p in inds], dtype=np.int64)
    b_idx = np.asarray([p[1] for p in inds], dtype=np.int64)
    traj = b.reshape(3, -1, b.shape[1], order="F")          # (3, atoms, cols)
    diff = traj[:, a_idx, :] - traj[:, b_idx, :]
    dist = np.sqrt(np.sum(diff * diff, axis=0))
    return dist.reshape((len(inds),) + s, order="F")


def flatpairdists(x: np.ndarray, cols: Optional[Sequence[int]] = None,
                  gram: bool = False) -> np.ndarray:
    """All-pairs distances in `halfinds` order.  src/utils/pairdists.jl:6-24.

    `cols` (0-based atom subset) is FeaturesAtoms (features.jl:26-29).
    gram=False: direct differences -- what the reference's own GPU kernel does
    (pairdists.jl:137-150).  gram=True: the reference's CPU path (:32-35), the
    cancellation-prone -2x'x + |x|^2 + |x|^2' followed by max(.,0) (:21); kept to
    show that the reference is not self-consistent to fp32 ulp."""
    b, s = _cols(x)
    c = b.shape[0] // 3
    traj = b.reshape(3, c, -1, order="F")
    if cols is not None:
        traj = traj[:, np.asarray(cols, dtype=np.int64), :]
        c = len(cols)
    inds = halfinds(c)
    ia = np.asarray([p[0] for p in inds], dtype=np.int64)
    ib = np.asarray([p[1] for p in inds], dtype=np.int64)
    if gram:
        sq = np.sum(traj * traj, axis=0)                    # (c, cols)
        g = np.einsum("kib,kjb->ijb", traj, traj)
        p = -2 * g + sq[:, None, :] + sq[None, :, :]
        p = p[ia, ib, :]
    else:
        diff = traj[:, ia, :] - traj[:, ib, :]
        p = np.sum(diff * diff, axis=0)
    p = np.sqrt(np.maximum(p, 0))
    return p.reshape((len(inds),) + s, order="F")


@dataclass
class FeaturesAll:
    """src/utils/features.jl:22-23"""
    def __call__(self, coords):
        return flatpairdists(coords)


@dataclass
class FeaturesAtoms:
    """src/utils/features.jl:26-29 (0-based atom indices here)"""
    atominds: Sequence[int]

    def __call__(self, coords):
        return flatpairdists(coords, self.atominds)


@dataclass
class FeaturesPairs:
    """src/utils/features.jl:31-34 (0-based pairs here)"""
    pairs: Sequence[Tuple[int, int]]

    def __call__(self, coords):
        return pdists(coords, self.pairs)


def identity(x):
    return x


# ----------------------------------------------------------------------------------
# model  (src/models.jl:59-92; Flux 0.16.9 Dense / LayerNorm / Chain semantics)
# ----------------------------------------------------------------------------------

ACTIVATIONS = ("identity", "sigmoid", "tanh", "relu", "swish")


def _act(name: str, z: np.ndarray) -> np.ndarray:
    if name == "identity":
        return z
    if name == "sigmoid":
        # NNlib sigmoid(_fast): 1/(1+exp(-x)), evaluated in the overflow-free form
        e = np.exp(-np.abs(z))
        return np.where(z >= 0, 1 / (1 + e), e / (1 + e))
    if name == "tanh":
        return np.tanh(z)
    if name == "relu":
        return np.maximum(z, 0)
    if name == "swish":
        return z * _act("sigmoid", z)
    raise ValueError(name)


def _dact(name: str, z: np.ndarray, a: np.ndarray) -> np.ndarray:
    """derivative of the activation given pre-activation z and activation a"""
    if name == "identity":
        return np.ones_like(z)
    if name == "sigmoid":
        return a * (1 - a)
    if name == "tanh":
        return 1 - a * a
    if name == "relu":
        return (z > 0).astype(z.dtype)
    if name == "swish":
        s = _act("sigmoid", z)
        return s + a * (1 - s)
    raise ValueError(name)


def pairnet_widths(n: int, layers: int = 3, nout: int = 1) -> List[int]:
    """src/models.jl:65-67: [round(Int, n^(l/layers)) for l in layers:-1:1] ++ [nout].
    Julia's round(Int, x) rounds half to even, as Python's round() does."""
    return [int(round(n ** (l / layers))) for l in range(layers, 0, -1)] + [nout]


@dataclass
class Model:
    """Flux.Chain([LayerNorm(n)], Dense..., Dense) as built by densenet
    (src/models.jl:87-92).  W[l] is (out, in); b[l] is (out,)."""
    W: List[np.ndarray]
    b: List[np.ndarray]
    ln_scale: Optional[np.ndarray] = None       # None <=> layernorm=false
    ln_bias: Optional[np.ndarray] = None
    activation: str = "sigmoid"
    lastactivation: str = "identity"
    ln_eps: float = 1e-5
    # Flux `normalise` eps placement; Flux >= 0.14 (incl. the pinned 0.16.9):
    # (x-mu)/sqrt(var + eps^2) with the biased variance.  "std_eps": (x-mu)/(std+eps)
    # (Flux <= 0.13).  "var_eps": (x-mu)/sqrt(var+eps) (PyTorch).  Source is not
    # under /root/reference, so the placement is an ASSUMPTION.
    ln_mode: str = "var_eps2"

    @property
    def widths(self) -> List[int]:
        return [self.W[0].shape[1]] + [w.shape[0] for w in self.W]

    @property
    def layernorm(self) -> bool:
        return self.ln_scale is not None

    def acts(self) -> List[str]:
        return [self.activation] * (len(self.W) - 1) + [self.lastactivation]

    def copy(self) -> "Model":
        return Model([w.copy() for w in self.W], [b.copy() for b in self.b],
                     None if self.ln_scale is None else self.ln_scale.copy(),
                     None if self.ln_bias is None else self.ln_bias.copy(),
                     self.activation, self.lastactivation, self.ln_eps, self.ln_mode)

    def astype(self, dt) -> "Model":
        m = self.copy()
        m.W = [w.astype(dt) for w in m.W]
        m.b = [b.astype(dt) for b in m.b]
        if m.layernorm:
            m.ln_scale = m.ln_scale.astype(dt)
            m.ln_bias = m.ln_bias.astype(dt)
        return m

    # ---- flat parameter vector in Flux trainables order -------------------------
    def arrays(self) -> List[np.ndarray]:
        out = []
        if self.layernorm:
            out += [self.ln_scale, self.ln_bias]
        for w, b in zip(self.W, self.b):
            out += [w, b]
        return out

    def get_flat(self) -> np.ndarray:
        return np.concatenate([a.reshape(-1, order="F") for a in self.arrays()])

    def set_flat(self, flat: np.ndarray) -> None:
        off = 0
        for a in self.arrays():
            n = a.size
            a[...] = np.asarray(flat[off:off + n]).reshape(a.shape, order="F")
            off += n
        assert off == len(flat)

    def nparams(self) -> int:
        return sum(a.size for a in self.arrays())

    # ---- forward ------------------------------------------------------------------
    def __call__(self, x: np.ndarray) -> np.ndarray:
        return forward(self, x)


def glorot_model(widths: Sequence[int], rng: np.random.Generator, *, layernorm: bool = True,
                 activation="sigmoid", lastactivation="identity", dtype=np.float64) -> Model:
    """densenet(layers=widths, ...) with Flux's default init: glorot_uniform weights
    U(+-sqrt(6/(in+out))), zero bias, LayerNorm scale 1 / bias 0.  Julia's RNG stream
    cannot be reproduced, so parity always INJECTS these weights on both sides."""
    W, b = [], []
    for i in range(len(widths) - 1):
        fin, fout = widths[i], widths[i + 1]
        lim = math.sqrt(6.0 / (fin + fout))
        W.append(rng.uniform(-lim, lim, size=(fout, fin)).astype(dtype))
        b.append(np.zeros(fout, dtype=dtype))
    ls = np.ones(widths[0], dtype=dtype) if layernorm else None
    lb = np.zeros(widths[0], dtype=dtype) if layernorm else None
    return Model(W, b, ls, lb, activation, lastactivation)


def pairnet(n: int, layers: int = 3, nout: int = 1, *, rng=None, layernorm=True,
            activation="sigmoid", lastactivation="identity", dtype=np.float64) -> Model:
    """src/models.jl:65-69"""
    rng = np.random.default_rng(0) if rng is None else rng
    return glorot_model(pairnet_widths(n, layers, nout), rng, layernorm=layernorm,
                        activation=activation, lastactivation=lastactivation, dtype=dtype)


def layernorm_stats(x2: np.ndarray, eps: float, mode: str):
    """per-column mean and 1/denominator over the feature axis (Flux `normalise`,
    dims=1, biased variance)."""
    mu = x2.mean(axis=0, keepdims=True)
    var = ((x2 - mu) ** 2).mean(axis=0, keepdims=True)
    if mode == "var_eps2":
        rstd = 1 / np.sqrt(var + eps * eps)
    elif mode == "std_eps":
        rstd = 1 / (np.sqrt(var) + eps)
    elif mode == "var_eps":
        rstd = 1 / np.sqrt(var + eps)
    else:
        raise ValueError(mode)
    return mu, rstd


def forward_cache(m: Model, x: np.ndarray):
    """model(x) for x::(F, cols...) keeping what backward needs.
    Flux.Dense flattens trailing dims; LayerNorm normalises over dim 1."""
    x2, s = _cols(x)
    cache = {"x": x2}
    a = x2
    if m.layernorm:
        mu, rstd = layernorm_stats(x2, m.ln_eps, m.ln_mode)
        xhat = (x2 - mu) * rstd
        a = xhat * m.ln_scale[:, None] + m.ln_bias[:, None]
        cache["xhat"] = xhat
    acts_in, zs = [], []
    for W, b, act in zip(m.W, m.b, m.acts()):
        acts_in.append(a)
        z = W @ a + b[:, None]
        zs.append(z)
        a = _act(act, z)
    cache["a_in"], cache["z"], cache["out"] = acts_in, zs, a
    return a.reshape((a.shape[0],) + s, order="F"), cache


def forward(m: Model, x: np.ndarray) -> np.ndarray:
    return forward_cache(m, x)[0]


def backward(m: Model, cache, dout: np.ndarray) -> List[np.ndarray]:
    """gradients wrt every trainable array, in Model.arrays() order, given
    dL/d(out)::(nout, B).  (What Zygote produces inside Flux.train!, src/iso.jl:184.)
    The gradient wrt the model input is not needed on the train path (features are
    leaf inputs) and is not formed."""
    acts = m.acts()
    gW, gb = [None] * len(m.W), [None] * len(m.W)
    da = dout
    a_out = cache["out"]
    for l in range(len(m.W) - 1, -1, -1):
        z = cache["z"][l]
        a_l = a_out if l == len(m.W) - 1 else cache["a_in"][l + 1]
        dz = da * _dact(acts[l], z, a_l)
        gW[l] = dz @ cache["a_in"][l].T
        gb[l] = dz.sum(axis=1)
        da = m.W[l].T @ dz
    grads = []
    if m.layernorm:
        grads += [(da * cache["xhat"]).sum(axis=1), da.sum(axis=1)]
    for w, b in zip(gW, gb):
        grads += [w, b]
    return grads


# ----------------------------------------------------------------------------------
# targets  (src/isotarget.jl:10-42, 74-127)
# ----------------------------------------------------------------------------------

def expectation(f, ys: np.ndarray) -> np.ndarray:
    """src/isotarget.jl:18: dropdims(sum(f(ys); dims=2); dims=2) ./ size(ys, 2)"""
    out = f(ys)
    return out.sum(axis=1) / ys.shape[1]


def shiftscale(ks: np.ndarray) -> np.ndarray:
    """src/isotarget.jl:36-42"""
    if not (ks.ndim == 1 or ks.shape[0] == 1):
        raise AssertionError("TransformShiftscale only works with one dimensional chi functions")
    lo, hi = ks.min(), ks.max()
    if not hi > lo:
        raise DomainError("Could not compute the shift-scale. chi function is constant")
    return (ks - lo) / (hi - lo)


def indexmap(X: np.ndarray) -> List[int]:
    """PCCAPlus.indexmap (v1.1.2; source not under /root/reference -- restated from
    the published inner-simplex algorithm, Weber & Galliat 2002): pick the rows of
    X::(N, d) spanning the largest simplex.  Step 1 takes the max-norm row and
    translates it to the origin; later steps take the max-norm row, normalise by
    its norm and project it out.  First maximal index on ties (Julia argmax)."""
    X = np.array(X, dtype=np.float64, copy=True)
    d = X.shape[1]
    ind = []
    for j in range(d):
        rownorm = np.sqrt(np.sum(X * X, axis=1))
        i = int(np.argmax(rownorm))
        ind.append(i)
        if j == 0:
            X = X - X[i:i + 1, :]
        else:
            X = X / rownorm[i]
            vt = X[i:i + 1, :]
            X = X - (X @ vt.T) @ vt
    return ind


def myisa(X: np.ndarray, whitening: bool = False) -> np.ndarray:
    """src/isotarget.jl:81-98: inv(X[indexmap(X), :]) in Float64; DomainError if singular."""
    X = np.asarray(X, dtype=np.float64)
    try:
        if whitening:
            C = X.T @ X / X.shape[0]
            ev, U = np.linalg.eigh(C)
            Wm = (U * ev ** -0.5) @ U.T
            i = indexmap(X @ Wm)
        else:
            i = indexmap(X)
        A = np.linalg.inv(X[i, :])
        if not np.all(np.isfinite(A)):
            raise np.linalg.LinAlgError("singular")
        return A
    except np.linalg.LinAlgError:
        raise DomainError("Could not compute the simplex transformation. "
                          "The subspace might be singular/collapsed")


def fixperm(new: np.ndarray, old: np.ndarray) -> np.ndarray:
    """src/isotarget.jl:120-127: argmin over all row permutations (lexicographic
    order, first minimiser) of the entrywise L1 distance to `old`."""
    return new[list(fixperm_index(new, old)), :]


def fixperm_index(new, old) -> Tuple[int, ...]:
    n = new.shape[0]
    if n > 8:
        # "TODO: use the hungarian algorithm for larger systems" (src/isotarget.jl:121): the reference's own search is n! steps.
        # cost(p) = sum_i sum_n |new[p_i, n] - old[i, n]| is a linear assignment problem on C[a, i] = sum_n |new[a, n] - old[i, n]|;
        # its minimiser is the brute force's wherever that is unique.  scipy's solver: an implementation independent of the product's.
        from scipy.optimize import linear_sum_assignment
        C = np.abs(new[:, None, :] - old[None, :, :]).sum(axis=2)          # C[a, i]
        rows, cols = linear_sum_assignment(C.T)                            # for every old row i (rows) a new row a (cols)
        return tuple(int(a) for a in cols[np.argsort(rows)])
    best, bestp = None, None
    for p in itertools.permutations(range(n)):
        c = np.abs(new[list(p), :] - old).sum()
        if best is None or c < best:
            best, bestp = c, p
    return bestp


@dataclass
class TransformShiftscale:
    """src/isotarget.jl:32-34"""
    pass


@dataclass
class TransformISA:
    """src/isotarget.jl:74-77"""
    permute: bool = True
    whitening: bool = False


@dataclass
class TransformPseudoInv:
    """src/isotarget.jl:145-150 (experimental; exercised by test/runtests.jl:59)"""
    normalize: bool = True
    direct: bool = True
    eigenvecs: bool = True
    permute: bool = True


def pseudoinv_target(t: "TransformPseudoInv", chi: np.ndarray, kchi: np.ndarray) -> np.ndarray:
    """src/isotarget.jl:152-179 from chi = model(xs) and kchi = E[model(ys)], both (d, N).
    The reference works in the model's Float32 here; the oracle keeps its fp64 convention.
    `schur(M).vectors` is LAPACK gees' Z (real Schur form) -- scipy.linalg.schur(output="real") calls the same routine."""
    import scipy.linalg
    chi = np.asarray(chi, np.float64)
    kchi = np.asarray(kchi, np.float64)
    if not chi.shape[0] > 1:
        raise AssertionError("TransformPseudoInv does not work with one dimensional chi functions")
    try:
        kchi_inv = np.linalg.pinv(kchi)                                    # :159-163
    except np.linalg.LinAlgError as err:
        raise DomainError("Could not compute the pseudoinverse. The subspace might be singular/collapsed") from err
    d = chi.shape[0]
    if t.direct:                                                           # :165-168
        Kinv = chi @ kchi_inv
        T = scipy.linalg.schur(Kinv, output="real")[1] if t.eigenvecs else np.eye(d)
        target = T @ Kinv @ kchi
    else:                                                                  # :169-173 ("this does not work at all!")
        K = kchi @ kchi_inv
        T = scipy.linalg.schur(K, output="real")[1] if t.eigenvecs else np.eye(d)
        target = T @ np.linalg.inv(K) @ kchi
    if t.normalize:                                                        # :175  rows ./ L1 norm .* N
        target = target / np.abs(target).sum(axis=1, keepdims=True) * target.shape[1]
    if t.permute:                                                          # :176
        target = fixperm(target, chi)
    return target


def isotarget(target, model: Model, xs: np.ndarray, ys: np.ndarray) -> np.ndarray:
    """src/isotarget.jl:12 dispatching to :34 (1-D) or :100-107 (ISA).
    xs/ys are FEATURES here (the reference passes features(data)/propfeatures(data))."""
    if isinstance(target, TransformShiftscale):
        return shiftscale(expectation(model, ys))
    if isinstance(target, TransformISA):
        chi = model(xs)
        if not chi.shape[0] > 1:
            raise AssertionError("TransformISA does not work with one dimensional chi functions")
        ks = expectation(model, ys)
        A = myisa(ks.T, target.whitening)
        t = A.T @ ks.astype(np.float64)
        if target.permute:
            t = fixperm(t, chi)
        return t.astype(xs.dtype)            # T(target): back to the feature eltype
    if isinstance(target, TransformPseudoInv):
        return pseudoinv_target(target, model(xs), expectation(model, ys)).astype(xs.dtype)
    raise TypeError(target)


# ----------------------------------------------------------------------------------
# optimisers  (src/models.jl:4-20; Optimisers.jl 0.4.7 rules)
# ----------------------------------------------------------------------------------

@dataclass
class OptRule:
    kind: str = "nesterov"        # "adam" | "nesterov"
    lr: float = 1e-3
    reg: float = 1e-4             # WeightDecay lambda, applied to EVERY trainable array
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    rho: float = 0.9


def AdamRegularized(adam=1e-3, reg=1e-4) -> OptRule:
    """src/models.jl:12"""
    return OptRule("adam", adam, reg)


def NesterovRegularized(lr=1e-3, reg=1e-4) -> OptRule:
    """src/models.jl:20 (the reference default, src/iso.jl:18)"""
    return OptRule("nesterov", lr, reg)


@dataclass
class OptState:
    """Flux.setup(rule, model) state: per-array Adam (m, v, beta^t) or Nesterov vel."""
    rule: OptRule
    m: List[np.ndarray] = field(default_factory=list)   # Adam mt / Nesterov velocity
    v: List[np.ndarray] = field(default_factory=list)   # Adam vt
    bt1: float = 0.9                                    # beta1^t, starts at beta1
    bt2: float = 0.999


def setup(rule: OptRule, model: Model) -> OptState:
    st = OptState(rule)
    st.m = [np.zeros_like(a) for a in model.arrays()]
    st.v = [np.zeros_like(a) for a in model.arrays()]
    st.bt1, st.bt2 = rule.beta1, rule.beta2
    return st


def update(st: OptState, model: Model, grads: List[np.ndarray]) -> None:
    """Optimisers.update! with OptimiserChain(WeightDecay(reg), rule):
    g <- g + reg*theta; then
    Adam:     m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2;
              theta <- theta - lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps); b^t <- b^t * b
    Nesterov: d = -rho^2 vel + (1+rho) lr g; vel <- rho vel - lr g; theta <- theta - d"""
    r = st.rule
    for k, (theta, g) in enumerate(zip(model.arrays(), grads)):
        g = g + r.reg * theta
        if r.kind == "adam":
            st.m[k] = r.beta1 * st.m[k] + (1 - r.beta1) * g
            st.v[k] = r.beta2 * st.v[k] + (1 - r.beta2) * g * g
            d = st.m[k] / (1 - st.bt1) / (np.sqrt(st.v[k] / (1 - st.bt2)) + r.eps) * r.lr
        elif r.kind == "nesterov":
            d = -r.rho * r.rho * st.m[k] + (1 + r.rho) * r.lr * g
            st.m[k] = r.rho * st.m[k] - r.lr * g
        else:
            raise ValueError(r.kind)
        theta -= d.astype(theta.dtype)
    if r.kind == "adam":
        st.bt1 *= r.beta1
        st.bt2 *= r.beta2


# ----------------------------------------------------------------------------------
# training  (src/iso.jl:72-94, 179-194)
# ----------------------------------------------------------------------------------

def batch_size(n: int, minibatch: int) -> int:
    """src/iso.jl:180"""
    return n if (minibatch == 0 or n < minibatch) else minibatch


def loss_weights(target: np.ndarray):
    """src/iso.jl:183: 1 ./ std(target, dims=2) (corrected, N-1) for nout > 1 else 1."""
    if target.shape[0] > 1:
        return 1.0 / np.std(target, axis=1, ddof=1)
    return np.ones(1, dtype=target.dtype)


def batch_loss_and_grads(model: Model, x: np.ndarray, y: np.ndarray, w: np.ndarray):
    """l = sum(abs2, (m(x) .- y) .* w); gradient of l / numobs(x)  (src/iso.jl:185,191)"""
    out, cache = forward_cache(model, x)
    r = (out - y) * w[:, None]
    l = float(np.sum(r * r))
    dout = 2 * r * w[:, None] / x.shape[1]
    return l, backward(model, cache, dout), cache


def train_batch(model: Model, xs: np.ndarray, target: np.ndarray, opt: OptState, minibatch: int,
                perm: Optional[np.ndarray] = None, rng: Optional[np.random.Generator] = None,
                partial: bool = False) -> float:
    """src/iso.jl:179-194.  `perm` (0-based permutation of 1:N) is the DataLoader's
    shuffle order; Julia's RNG stream cannot be reproduced so parity injects it.
    partial=false drops the ragged tail; the returned loss still divides by the
    full N (src/iso.jl:193)."""
    n = xs.shape[1]
    bs = batch_size(n, minibatch)
    if perm is None:
        perm = (rng or np.random.default_rng()).permutation(n)
    w = loss_weights(target)
    ls = 0.0
    nb = n // bs if not partial else -(-n // bs)
    for b in range(nb):
        idx = perm[b * bs:(b + 1) * bs]
        l, grads, _ = batch_loss_and_grads(model, xs[:, idx], target[:, idx], w)
        if not math.isfinite(l):
            raise DomainError("The ISOKANN model collapsed under training. "
                              "Try reducing the learning rate or increasing regularization")
        ls += l
        update(opt, model, grads)
    return ls / n


@dataclass
class SimulationData:
    """src/simulation.jl:71-76, 100-114: coords=(xs, ys), features=(f(xs), f(ys)) cast
    to Float32 in the reference (kept in `dtype` here, float64 by default so the
    oracle is the tighter side of every comparison)."""
    xs: np.ndarray
    ys: np.ndarray
    featurizer: object = identity
    dtype: object = np.float64

    def __post_init__(self):
        self.features = (np.asarray(self.featurizer(self.xs), dtype=self.dtype),
                         np.asarray(self.featurizer(self.ys), dtype=self.dtype))

    def __len__(self):
        return self.features[0].shape[1]


@dataclass
class Iso:
    """src/iso.jl:3-43"""
    data: SimulationData
    model: Model
    opt: OptState
    target: object = None
    minibatch: int = 100
    losses: List[float] = field(default_factory=list)

    def __post_init__(self):
        if self.target is None:                       # src/iso.jl:28-34
            self.target = TransformShiftscale() if self.model.widths[-1] == 1 else TransformISA()


def run(iso: Iso, n: int = 1, epochs: int = 1, perms: Optional[Sequence[np.ndarray]] = None,
        rng: Optional[np.random.Generator] = None) -> Iso:
    """run!(iso, n, epochs)  src/iso.jl:72-94.  perms: one permutation per
    (iteration, epoch), consumed in order."""
    k = 0
    for _ in range(n):
        target = isotarget(iso.target, iso.model, iso.data.features[0], iso.data.features[1])
        xs = iso.data.features[0]
        for _ in range(epochs):
            perm = None if perms is None else perms[k]
            k += 1
            loss = train_batch(iso.model, xs, target, iso.opt, iso.minibatch, perm=perm, rng=rng)
            iso.losses.append(loss)
    return iso


def chis(iso: Iso) -> np.ndarray:
    """src/iso.jl:203"""
    return iso.model(iso.data.features[0])


def koopman(iso: Iso) -> np.ndarray:
    """src/isotarget.jl:20"""
    return expectation(iso.model, iso.data.features[1])


def validationloss(iso: Iso, valdata: SimulationData) -> float:
    """src/iso.jl:160-168: mean(abs2, chi(vx) - shiftscale([K chi(vy); K chi(ys)])[1:nv]); 1-D chi"""
    xs, ys = iso.data.features
    vx, vy = valdata.features
    c = iso.model(vx).reshape(-1)
    k1 = expectation(iso.model, vy).reshape(-1)
    k2 = expectation(iso.model, ys).reshape(-1)
    skc = shiftscale(np.concatenate([k1, k2]))[:len(c)]
    return float(np.mean((c - skc) ** 2))


def koopman_variance(iso: Iso, ys_features: Optional[np.ndarray] = None) -> float:
    """src/iso.jl:355-359: sum(abs2, chi - mean_k chi) / nout / N over the propagated samples"""
    ys = iso.data.features[1] if ys_features is None else ys_features
    chi = iso.model(ys)
    i, k, n = chi.shape
    return float(np.sum((chi - chi.sum(axis=1, keepdims=True) / k) ** 2) / i / n)


def rates(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """src/iso.jl:345-351: log(y / x) with the 1-D case extended by (1 - chi); y / x is Julia's
    right division (least squares for non-square operands)"""
    import scipy.linalg
    if x.shape[0] == 1:
        x = np.concatenate([x, 1 - x], axis=0)
        y = np.concatenate([y, 1 - y], axis=0)
    q = np.linalg.lstsq(x.T, y.T, rcond=None)[0].T
    return scipy.linalg.logm(q)


# ----------------------------------------------------------------------------------
# synthetic workloads shared by tests, smoke and bench
# ----------------------------------------------------------------------------------

def synthetic_coords(n_atoms: int, n: int, k: int, seed: int = 0, dtype=np.float32):
    """random-walk polymer, per-sample jitter 0.05 nm for xs, extra 0.02 nm per
    replica for ys.  Returns xs::(3A, N), ys::(3A, K, N) in Julia layout
    (Fortran-ordered so a sample's 3A values are contiguous)."""
    rng = np.random.default_rng(seed)
    u = rng.normal(size=(n_atoms, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    base = np.cumsum(0.15 * u, axis=0).reshape(-1)                 # (3A,)
    xs = base[None, :] + 0.05 * rng.standard_normal((n, 3 * n_atoms), dtype=np.float32)
    ys = xs[:, None, :] + 0.02 * rng.standard_normal((n, k, 3 * n_atoms), dtype=np.float32)
    xs = np.ascontiguousarray(xs, dtype=dtype).T                  # (3A, N), F-ordered view
    ys = np.ascontiguousarray(ys, dtype=dtype).transpose(2, 1, 0)  # (3A, K, N)
    return xs, ys


def random_pairs(n_atoms: int, nfeat: int, seed: int = 0) -> List[Tuple[int, int]]:
    """nfeat distinct sorted pairs i<j (mirrors FeaturesPairs(...; maxfeatures),
    src/utils/features.jl:60-62)."""
    rng = np.random.default_rng(seed + 12345)
    total = n_atoms * (n_atoms - 1) // 2
    sel = np.sort(rng.choice(total, size=nfeat, replace=False))
    # unrank in halfinds order: pair index p -> (i, j), j(j-1)/2 <= p
    j = np.floor((1 + np.sqrt(1 + 8 * sel.astype(np.float64))) / 2).astype(np.int64)
    j = np.where(j * (j - 1) // 2 > sel, j - 1, j)
    i = sel - j * (j - 1) // 2
    pairs = sorted((int(a), int(b)) for a, b in zip(i, j))
    return pairs
