"""Generates the golden vectors tests/golden/*.npz from the CPU oracle (fp64).

The reference itself cannot run here (no Julia), so these are
oracle-generated fixtures: inputs (coords, pair list, injected initial weights,
injected shuffle permutations) and expected outputs (first target, per-epoch
losses, final chi values, final parameters).  Run from the repo root:

    python -m tests.golden.make_golden
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import isokann_oracle as O  # noqa: E402

CASES = {
    # name: (n_atoms or None, nfeat pairs or "all" or None, D, N, K, nout, opt, minibatch, n_iter)
    "c1_shiftscale_nesterov": dict(atoms=None, feat=None, D=2, N=100, K=5, nout=1,
                                   opt=("nesterov", 1e-3, 1e-4), minibatch=100, n_iter=5),
    "c2small_shiftscale_adam": dict(atoms=8, feat="all", N=64, K=4, nout=1,
                                    opt=("adam", 1e-3, 1e-4), minibatch=16, n_iter=4),
    "c3small_isa_adam": dict(atoms=20, feat=40, N=96, K=5, nout=3,
                             opt=("adam", 1e-3, 1e-4), minibatch=32, n_iter=4),
}


def make_inputs(name):
    c = CASES[name]
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    if c["atoms"] is None:
        xs = rng.uniform(size=(c["D"], c["N"]))
        ys = rng.uniform(size=(c["D"], c["K"], c["N"]))
        pairs = np.zeros((0, 2), dtype=np.int32)
        kind = 0
        F = c["D"]
    else:
        xs, ys = O.synthetic_coords(c["atoms"], c["N"], c["K"], seed=3, dtype=np.float64)
        xs, ys = np.array(xs), np.array(ys)
        if c["feat"] == "all":
            pairs = np.zeros((0, 2), dtype=np.int32)
            kind = 1
            F = c["atoms"] * (c["atoms"] - 1) // 2
        else:
            pairs = np.asarray(O.random_pairs(c["atoms"], c["feat"], seed=1), dtype=np.int32)
            kind = 2
            F = c["feat"]
    widths = O.pairnet_widths(F, nout=c["nout"])
    m = O.glorot_model(widths, rng)
    # non-trivial LN affine and biases so every parameter class is exercised
    m.ln_scale[...] = 1 + 0.05 * rng.normal(size=F)
    m.ln_bias[...] = 0.05 * rng.normal(size=F)
    for b in m.b:
        b[...] = 0.05 * rng.normal(size=b.shape)
    perms = np.stack([rng.permutation(c["N"]) for _ in range(c["n_iter"])]).astype(np.int32)
    return dict(xs=xs, ys=ys, pairs=pairs, featurizer_kind=np.int32(kind), widths=np.asarray(widths, np.int32),
                params0=m.get_flat(), perms=perms, minibatch=np.int32(c["minibatch"]),
                n_iter=np.int32(c["n_iter"]), opt_kind=np.array(c["opt"][0]), opt_lr=np.float64(c["opt"][1]),
                opt_reg=np.float64(c["opt"][2]))


def build_case(name, g):
    """rebuild the oracle Iso from a golden file's inputs"""
    kind = int(g["featurizer_kind"])
    feat = {0: O.identity, 1: O.FeaturesAll(), 2: O.FeaturesPairs([tuple(p) for p in g["pairs"]])}[kind]
    data = O.SimulationData(g["xs"], g["ys"], feat)
    widths = [int(w) for w in g["widths"]]
    m = O.glorot_model(widths, np.random.default_rng(0))
    m.set_flat(g["params0"])
    rule = O.OptRule(str(g["opt_kind"]), float(g["opt_lr"]), float(g["opt_reg"]))
    iso = O.Iso(data, m, O.setup(rule, m), minibatch=int(g["minibatch"]))
    return iso, [p.astype(np.int64) for p in g["perms"]]


def main():
    for name in CASES:
        g = make_inputs(name)
        iso, perms = build_case(name, g)
        g["target0"] = O.isotarget(iso.target, iso.model, *iso.data.features)
        O.run(iso, int(g["n_iter"]), 1, perms=perms)
        g["losses"] = np.asarray(iso.losses)
        g["chis"] = O.chis(iso)
        g["koopman"] = O.koopman(iso)
        g["params_final"] = iso.model.get_flat()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **g)
        print(name, "losses", iso.losses, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
