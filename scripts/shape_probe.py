"""GPU: shapes at the EDGES of what isk_create accepts (scripts/fuzz_parity.py draws from the middle): very wide / very narrow feature rows on
both featurizers, atom counts around the register featurizer's limits, hidden widths around every kernel-selection boundary, deep
networks, many outputs, tiny N / K, every forward arithmetic.  Per case: chi against the oracle, one epoch of train_batch! (loss), the
coordinate pullback where it applies (finite, non-zero).  Prints one line per case; exit code 1 if any case fails.
    python scripts/shape_probe.py"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import isokann_oracle as O  # noqa: E402

pkg = ge.load_package()


def mk_model(om):
    m = pkg.densenet(om.widths, activation=om.activation, lastactivation=om.lastactivation, layernorm=om.layernorm)
    m.ln_mode = om.ln_mode
    m.params = om.get_flat().astype(np.float32)
    return m


def rand_model(widths, seed, ln=True):
    rng = np.random.default_rng(seed)
    om = O.glorot_model(widths, rng, layernorm=ln)
    for b in om.b:
        b[...] = 0.1 * rng.normal(size=b.shape)
    if ln:
        om.ln_scale[...] = 1 + 0.1 * rng.normal(size=widths[0])
        om.ln_bias[...] = 0.1 * rng.normal(size=widths[0])
    return om


CASES = []
for F in (4, 5, 31, 33, 2047, 2048, 2049, 5000, 10000):
    CASES.append(dict(kind="identity", F=F, tail=[20, 1]))
for A in (2, 3, 4, 64, 65, 100):
    CASES.append(dict(kind="all", A=A, tail=[12, 2 if A > 3 else 1]))
for A, F in ((682, 300), (683, 300), (2000, 100), (3000, 1000), (70, 2049), (300, 4000)):
    CASES.append(dict(kind="pairs", A=A, F=F, tail=[24, 3]))
for tail in ([1, 1], [160, 1], [161, 1], [160, 16, 1], [160, 17, 1], [100, 100, 100, 1], [512, 512, 512, 1], [513, 2], [1000, 16, 16, 16, 16, 16, 1],
             [30, 20, 64], [30, 9], [2000, 1200, 3]):
    CASES.append(dict(kind="identity", F=48, tail=tail))
for N, K, mb in ((2, 1, 0), (3, 3, 2), (15, 2, 16), (17, 1, 1), (33, 5, 32)):
    CASES.append(dict(kind="identity", F=20, tail=[8, 1], N=N, K=K, mb=mb))


def run_case(c, mode):
    rng = np.random.default_rng(7)
    N, K, mb = c.get("N", 64), c.get("K", 2), c.get("mb", 32)
    if c["kind"] == "identity":
        F = c["F"]
        xs, ys = rng.normal(size=(F, N)), rng.normal(size=(F, K, N))
        data, feat = pkg.SimulationData(xs, ys, pkg.identity), None
    else:
        A = c["A"]
        xs, ys = O.synthetic_coords(A, N, K, seed=3)
        if c["kind"] == "all":
            F = A * (A - 1) // 2
            data, feat = pkg.SimulationData(xs, ys, pkg.FeaturesAll()), O.FeaturesAll()
        else:
            F = c["F"]
            pairs = O.random_pairs(A, F, seed=4)
            data, feat = pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), O.FeaturesPairs(pairs)
    widths = [F] + list(c["tail"])
    nout = widths[-1]
    if nout > 1 and N < 2 * nout + 2:
        return "skipped (fewer points than an ISA simplex needs)"
    om = rand_model(widths, seed=5, ln=F >= 4)
    odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64), feat) if feat is not None else \
        O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64))
    iso = pkg.Iso(data, model=mk_model(om), opt=pkg.AdamRegularized(1e-3, 1e-4), minibatch=mb, forward_mode=mode,
                  target=pkg.TransformISA() if nout > 1 else None)
    try:
        want = O.forward(om, odata.features[0])
        got = pkg.chis(iso)
        tol = 3e-2 if mode == 2 else 1e-4
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        assert err <= tol, f"chi err {err:.2e}"
        tgt = np.asarray(rng.uniform(size=(nout, N)), np.float32)
        st = O.setup(O.AdamRegularized(1e-3, 1e-4), om)
        perm = rng.permutation(N)
        lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), st, mb, perm=perm)
        l = pkg.train_batch(iso, tgt, perm=perm)
        assert abs(l - lo) <= 1e-3 * abs(lo) + 1e-12, f"loss {l} vs {lo}"
        g = pkg.chicoords_pullback(iso, xs[:, :2], np.ones((nout, 2)))
        assert g.shape == (xs.shape[0], 2) and np.all(np.isfinite(g)), "pullback not finite"
        return f"ok (chi err {err:.1e})"
    finally:
        iso.close()


def main():
    fails = 0
    for c in CASES:
        for mode in (0, 1, 2, 3):
            tag = f"{c} mode {mode}"
            try:
                r = run_case(c, mode)
                if r is None:
                    continue
                print("PASS" if r.startswith("ok") else "SKIP", tag, r, flush=True)
            except Exception as ex:  # noqa: BLE001
                fails += 1
                print("FAIL", tag, repr(ex)[:300], flush=True)
                if os.environ.get("PROBE_TRACE"):
                    traceback.print_exc()
    print(f"{fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
