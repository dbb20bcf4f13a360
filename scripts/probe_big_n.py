"""GPU: large-N shapes against the oracle (tests/test_gpu_big_n.py)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import __graft_entry__ as ge
from oracle import isokann_oracle as O
pkg = ge.load_package()
from shape_probe import rand_model, mk_model


def main():
    fails = 0
    for N, K, nout, mb in ((70000, 1, 3, 1000), (300000, 2, 3, 0), (300000, 1, 4, 4096), (1200000, 1, 2, 100000), (250001, 3, 1, 7)):
        rng = np.random.default_rng(N)
        F = 8
        xs, ys = rng.normal(size=(F, N)).astype(np.float32), rng.normal(size=(F, K, N)).astype(np.float32)
        om = rand_model([F, 16, nout], seed=2)
        odata = O.SimulationData(xs.astype(np.float64), ys.astype(np.float64))
        tr = pkg.TransformISA() if nout > 1 else None
        otr = O.TransformISA() if nout > 1 else O.TransformShiftscale()
        iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.identity), model=mk_model(om), target=tr, minibatch=mb, opt=pkg.AdamRegularized(1e-3, 1e-4))
        try:
            t = pkg.isotarget(iso)
            want = O.isotarget(otr, om, *odata.features)
            err = np.abs(t - want).max() / np.abs(want).max()
            perm = rng.permutation(N)
            st = O.setup(O.AdamRegularized(1e-3, 1e-4), om)
            tgt = np.asarray(rng.uniform(size=(nout, N)), np.float32)
            if mb and N // mb > 300:
                l, lo = 0, 0
            else:
                lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), st, mb, perm=perm)
                l = pkg.train_batch(iso, tgt, perm=perm)
            ok = err < 2e-4 and abs(l - lo) <= 2e-4 * abs(lo) + 1e-12
            print("PASS" if ok else "FAIL", N, K, nout, mb, "target err", err, "loss", l, lo, flush=True)
            fails += not ok
        except Exception as ex:
            fails += 1
            print("FAIL", N, K, nout, mb, repr(ex)[:300], flush=True)
        finally:
            iso.close()
    print(fails, "failures")
    return fails


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
