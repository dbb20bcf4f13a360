"""GPU: random SEQUENCES of API calls against the oracle replaying the same sequence.  The engine caches derived state everywhere
(folded / split weight images and their dirty flags, the feature buffers of the next iteration written ahead on the side stream, epoch
graphs keyed by buffer, the device copy of beta^t, pre-gathered batch images): what the parametrized tests and scripts/fuzz_parity.py
exercise is one call pattern per test; this draws the pattern.  Operations: run (1-2 iterations), chis, koopman, isotarget,
train_batch on a random target, batch_grads, set_params (both sides), optimizer-state round trip, a changed minibatch, chicoords.
    python scripts/api_fuzz.py [nseq] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import isokann_oracle as O  # noqa: E402

pkg = ge.load_package()
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from shape_probe import mk_model, rand_model  # noqa: E402


def isa_undetermined(oiso, om, odata):
    """True where the ISA target of the oracle's current state is not determined by the data to Float32 accuracy: an ill-conditioned
    simplex (amplifies the forward's 1e-7), or two row permutations of fixperm at the same cost to rounding (deep sigmoid networks at
    their initialisation put K chi almost on a line: for two outputs both permutations then cost exactly the same and rounding noise
    picks one -- in the reference as well)"""
    import itertools
    ks = O.koopman(oiso).astype(np.float64)
    chi = O.chis(oiso).astype(np.float64)
    try:
        A = O.myisa(ks.T)
    except Exception:  # noqa: BLE001
        return True
    if not np.all(np.isfinite(A)) or np.abs(A).sum(axis=0).max() > 300:
        return True
    T = A.T @ ks
    d = T.shape[0]
    if d <= 4:
        costs = sorted(np.abs(T[list(p), :] - chi).sum() for p in itertools.permutations(range(d)))
        if len(costs) > 1 and costs[1] - costs[0] <= 1e-6 * max(costs[0], 1e-30):
            return True
    # the simplex vertices: a near-tie of the largest norms flips the choice
    return False


def one_sequence(rng, verbose, bf16=False):
    big = bool(os.environ.get("FUZZ_BIG"))            # larger problems: N to 3000, rows to 300 features, hidden widths to 400 (beyond 160: the layered path)
    kind = rng.choice(["identity", "pairs"])
    N = int(rng.integers(40, 3000 if big else 300))
    K = int(rng.integers(1, 5))
    nout = int(rng.choice([1, 1, 2, 3]))
    every = bool(rng.integers(0, 2))
    mode = 2 if bf16 else int(rng.choice([0, 1, 3]))
    layered = bool(rng.integers(0, 3) == 0) or (bf16 and bool(rng.integers(0, 2)))
    optk = str(rng.choice(["adam", "nesterov"]))
    if kind == "identity":
        F = int(rng.integers(6, 300 if big else 80))
        xs, ys = rng.normal(size=(F, N)), rng.normal(size=(F, K, N))
        feat_g, feat_o = pkg.identity, None
    else:
        A = int(rng.integers(8, 80 if big else 40))
        F = int(rng.integers(6, min(300 if big else 120, A * (A - 1) // 2)))
        xs, ys = O.synthetic_coords(A, N, K, seed=int(rng.integers(1 << 30)))
        pairs = O.random_pairs(A, F, seed=int(rng.integers(1 << 30)))
        feat_g, feat_o = pkg.FeaturesPairs(pairs), O.FeaturesPairs(pairs)
    hidden = [int(rng.integers(3, (400 if big else 200) if layered else (160 if big else 60))) for _ in range(int(rng.integers(1, 4)))]
    widths = [F] + hidden + [nout]
    om = rand_model(widths, seed=int(rng.integers(1 << 30)))
    mb = int(rng.choice([0, 16, 33, 64]))
    train16 = bf16 and bool(rng.integers(0, 2))
    # bf16 flavour (plain-bf16 forward, optionally bf16 training GEMMs): bf16 tolerances -- still far below what a stale weight image costs
    ftol, ltol = (4e-2, 6e-2) if bf16 else (5e-4, 5e-3)
    desc = dict(kind=kind, N=N, K=K, widths=widths, every=every, mode=mode, layered=layered, opt=optk, mb=mb, train16=train16)
    odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64), feat_o) if feat_o is not None else \
        O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64))
    lr, reg = 1e-3, 1e-4
    orule = O.AdamRegularized(lr, reg) if optk == "adam" else O.NesterovRegularized(lr, reg)
    grule = pkg.AdamRegularized(lr, reg) if optk == "adam" else pkg.NesterovRegularized(lr, reg)
    otr = O.TransformISA() if nout > 1 else O.TransformShiftscale()
    gtr = pkg.TransformISA() if nout > 1 else None
    oiso = O.Iso(odata, om, O.setup(orule, om), minibatch=mb, target=otr)
    iso = pkg.Iso(pkg.SimulationData(xs, ys, feat_g), model=mk_model(om), opt=grule, minibatch=mb, target=gtr, forward_mode=mode,
                  featurize_every_iteration=every, force_layered=layered, options={"train_bf16": 1} if train16 else None)
    log = []
    try:
        for step in range(int(rng.integers(4, 9))):
            op = str(rng.choice(["run", "run", "chis", "koopman", "target", "train", "grads", "setparams", "optstate", "minibatch", "chicoords", "append", "saveload"]))
            if bf16 and op in ("run", "target", "grads"):          # (trajectories and the ISA solve amplify bf16 rounding: forward / one-epoch ops only)
                op = str(rng.choice(["chis", "koopman", "train", "setparams", "chicoords"]))
            log.append(op)
            if op == "run":
                n = 1                                           # (one iteration per op: the simplex is checked in front of each)
                if nout > 1 and isa_undetermined(oiso, om, odata):
                    return "skip", desc, log
                perms = np.stack([rng.permutation(N) for _ in range(n)]).astype(np.int32)
                try:
                    O.run(oiso, n, 1, perms=[p.astype(np.int64) for p in perms])
                except O.DomainError:
                    return "skip", desc, log          # the oracle's own trajectory ends in a DomainError: conditioning, not arithmetic
                pkg.run(iso, n, 1, perms=perms)
                a, b = np.asarray(iso.losses[-n:]), np.asarray(oiso.losses[-n:])
                if not np.all(np.isfinite(b)) or np.abs(om.get_flat()).max() > 1e3:
                    return "skip", desc, log
                assert np.all(np.abs(a - b) <= 5e-3 * np.abs(b) + 1e-9), ("run losses", a, b)
            elif op == "chis":
                assert np.abs(pkg.chis(iso) - O.chis(oiso)).max() <= ftol * max(1.0, np.abs(O.chis(oiso)).max()), ("chis", np.abs(pkg.chis(iso) - O.chis(oiso)).max())
            elif op == "koopman":
                assert np.abs(pkg.koopman(iso) - O.koopman(oiso)).max() <= ftol * max(1.0, np.abs(O.koopman(oiso)).max()), "koopman"
            elif op == "target":
                try:
                    want = O.isotarget(otr, om, *odata.features)
                except O.DomainError:
                    return "skip", desc, log
                if nout > 1 and isa_undetermined(oiso, om, odata):
                    continue
                got = pkg.isotarget(iso)
                assert np.abs(got - want).max() <= 2e-3 * max(1.0, np.abs(want).max()), ("target", np.abs(got - want).max())
            elif op == "train":
                tgt = np.asarray(rng.uniform(size=(nout, N)), np.float32)
                perm = rng.permutation(N)
                lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), oiso.opt, mb, perm=perm)
                l = pkg.train_batch(iso, tgt, perm=perm)
                if not np.isfinite(lo) or np.abs(om.get_flat()).max() > 1e3:
                    return "skip", desc, log
                assert abs(l - lo) <= ltol * abs(lo) + 1e-9, ("train loss", l, lo)
                if bf16:                                        # keep the two sides together: the oracle continues from the ENGINE's parameters
                    om.set_flat(iso.model.get_params().astype(np.float64))
            elif op == "grads":
                tgt = np.asarray(rng.uniform(size=(nout, N)), np.float32)
                idx = rng.permutation(N)[: max(2, N // 3)]
                g, l = pkg.batch_grads(iso, tgt, idx.astype(np.int32))
                w = O.loss_weights(tgt.astype(np.float64))
                lo, go, _ = O.batch_loss_and_grads(om, odata.features[0][:, idx], tgt.astype(np.float64)[:, idx], w)
                assert abs(l - lo) <= 2e-3 * abs(lo) + 1e-9, ("grads loss", l, lo)
            elif op == "setparams":
                p = om.get_flat() + 0.05 * rng.normal(size=om.get_flat().shape)
                om.set_flat(p.astype(np.float32).astype(np.float64))
                iso.model.set_params(p.astype(np.float32))
            elif op == "optstate":
                m, v, bt = iso.get_opt_state()
                iso.set_opt_state(m, v, bt)
            elif op == "minibatch":
                mb = int(rng.choice([0, 16, 33, 64]))
                iso.minibatch = mb
                oiso.minibatch = mb
            elif op == "append":
                # addcoords! / the cutoff window (src/iso.jl:238,288-290): drop start points at the front, append new ones
                n_add = int(rng.integers(0, 20))
                drop = int(rng.integers(0, max(1, min(10, N - 30))))
                if kind == "identity":
                    xn, yn = rng.normal(size=(F, n_add)), rng.normal(size=(F, K, n_add))
                else:
                    xn, yn = O.synthetic_coords(A, max(n_add, 1), K, seed=int(rng.integers(1 << 30)))
                    xn, yn = xn[:, :n_add], yn[:, :, :n_add]
                if n_add == 0 and drop == 0:
                    continue
                pkg.append_data(iso, xn if n_add else None, yn if n_add else None, drop_first=drop)
                xs = np.concatenate([np.asarray(xs)[:, drop:], xn], axis=1) if n_add else np.asarray(xs)[:, drop:]
                ys = np.concatenate([np.asarray(ys)[:, :, drop:], yn], axis=2) if n_add else np.asarray(ys)[:, :, drop:]
                N = xs.shape[1]
                odata = O.SimulationData(np.asarray(xs, np.float32).astype(np.float64), np.asarray(ys, np.float32).astype(np.float64), feat_o) if feat_o is not None else \
                    O.SimulationData(np.asarray(xs, np.float32).astype(np.float64), np.asarray(ys, np.float32).astype(np.float64))
                oiso.data = odata
            elif op == "saveload":
                # save / load (src/iso.jl:405-420): parameters, optimizer state, data and losses through one file, a NEW engine
                path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"isk_api_fuzz_{os.getpid()}.npz")
                pkg.save(path, iso)
                iso.close()
                iso = pkg.load(path, force_layered=layered, options={"train_bf16": 1} if train16 else None)
                os.remove(path)
            elif op == "chicoords":
                x = xs[:, : min(7, N)]
                want = O.forward(om, odata.featurizer(np.asarray(x, np.float64)) if feat_o is not None else np.asarray(x, np.float64))
                assert np.abs(pkg.chicoords(iso, x) - want).max() <= ftol * max(1.0, np.abs(want).max()), "chicoords"
        return "ok", desc, log
    except AssertionError as ex:
        raise AssertionError((ex.args, desc, log)) from None
    finally:
        iso.close()


def sweep(nseq, seed, verbose=True, bf16=False):
    rng = np.random.default_rng(seed)
    fails, skips, failed = 0, 0, []
    for i in range(nseq):
        sub = np.random.default_rng(rng.integers(1 << 62))
        try:
            r, desc, log = one_sequence(sub, verbose, bf16)
            skips += r == "skip"
        except AssertionError as ex:
            fails += 1
            failed.append((i, repr(ex)[:400]))
            print("FAIL", i, repr(ex)[:600], flush=True)
        except Exception as ex:  # noqa: BLE001
            fails += 1
            failed.append((i, repr(ex)[:400]))
            print("ERROR", i, repr(ex)[:600], flush=True)
    print(f"{nseq} sequences, {skips} skipped, {fails} failures", flush=True)
    return fails, skips, failed


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if sweep(n, s, bf16=bool(os.environ.get("FUZZ_BF16")))[0] else 0)
