"""Synthetic workloads for bench.py / smoke: random-walk polymer
coordinates and random sorted atom pairs.  Input generation only -- no hot-path compute."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def synthetic_coords(n_atoms: int, n: int, k: int, seed: int = 0):
    """xs (3A, N), ys (3A, K, N) float32, Julia layout (a sample's 3A values contiguous).
    Atom a sits at sum_{i<=a} 0.15 u_i nm; per-sample jitter 0.05 nm for xs, extra 0.02 nm per
    replica for ys."""
    rng = np.random.default_rng(seed)
    u = rng.normal(size=(n_atoms, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    base = np.cumsum(0.15 * u, axis=0).reshape(-1).astype(np.float32)
    xs = base[None, :] + np.float32(0.05) * rng.standard_normal((n, 3 * n_atoms), dtype=np.float32)
    ys = np.empty((n, k, 3 * n_atoms), dtype=np.float32)
    for j in range(k):                       # chunked: keeps the peak host footprint at one ys + one slab
        ys[:, j, :] = xs + np.float32(0.02) * rng.standard_normal((n, 3 * n_atoms), dtype=np.float32)
    return xs.T, ys.transpose(2, 1, 0)      # F-ordered views


def random_pairs(n_atoms: int, nfeat: int, seed: int = 0) -> List[Tuple[int, int]]:
    """nfeat distinct sorted pairs i<j, as FeaturesPairs(...; maxfeatures) samples them
    (src/utils/features.jl:60-62 of the reference)"""
    rng = np.random.default_rng(seed + 12345)
    total = n_atoms * (n_atoms - 1) // 2
    sel = np.sort(rng.choice(total, size=nfeat, replace=False))
    j = np.floor((1 + np.sqrt(1 + 8 * sel.astype(np.float64))) / 2).astype(np.int64)
    j = np.where(j * (j - 1) // 2 > sel, j - 1, j)
    i = sel - j * (j - 1) // 2
    return sorted((int(a), int(b)) for a, b in zip(i, j))


WORKLOADS = {
    # name: atoms, features ("all" or count), N, K, nout   (BASELINE.json configs[1], configs[2])
    "c2": dict(atoms=22, feat="all", N=1000, K=10, nout=1, label="C2 ADP 22 atoms -> 231 all-pair feats, N=1000, K=10, [231,38,6,1]+LN, shiftscale"),
    "c3": dict(atoms=591, feat=1700, N=10000, K=20, nout=3, label="C3 villin 591 atoms -> 1700 pair feats, N=10000, K=20, [1700,142,12,3]+LN, TransformISA"),
    # one GPU's shard of BASELINE.json configs[4] (N = 100 000 over 8 GPUs): wide densenet without LayerNorm
    "c5shard": dict(atoms=304, feat=2000, N=12500, K=32, nout=1, widths=[2000, 512, 512, 512, 1], layernorm=False,
                    label="C5 shard: Trp-cage 304 atoms -> 2000 pair feats, N=12500 (=100000/8), K=32, densenet [2000,512,512,512,1], shiftscale"),
}
