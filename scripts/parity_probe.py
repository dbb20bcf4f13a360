"""Diagnostic (GPU): where does the C3-width run deviate from the fp64 oracle?  Teacher forcing: at every iteration the
engine's parameters and optimizer state are reset to the oracle's, then each stage is compared on its own --
forward (chi, K chi), ISA target, one full-batch epoch with the ORACLE's target.  python scripts/parity_probe.py [fwd_mode]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from oracle import isokann_oracle as O  # noqa: E402
import benchshape_cases as tb  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
opt = sys.argv[2] if len(sys.argv) > 2 else "adam"
pkg = ge.load_package()
N, K = 1000, 20
xs, ys, pairs = tb.coords(tb.C3_ATOMS, tb.C3_F, N, K, 9)
fx, fy = tb.features64(tb.C3_ATOMS, tb.C3_F, N, K, 9)
om = tb.model(tb.C3_WIDTHS, seed=13, ln=True)
rule = O.AdamRegularized() if opt == "adam" else O.NesterovRegularized()
st = O.setup(rule, om)
model = pkg.densenet(tb.C3_WIDTHS, layernorm=True)
model.params = om.get_flat().astype(np.float32)
popt = pkg.AdamRegularized() if opt == "adam" else pkg.NesterovRegularized()
iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=model, opt=popt, minibatch=0, forward_mode=mode)
f = pkg.features(iso)
print("features: max rel err", np.max(np.abs(f - fx) / np.abs(fx)))
for it in range(6):
    iso.model.set_params(om.get_flat().astype(np.float32))
    flat = lambda arrs: np.concatenate([a.reshape(-1, order="F") for a in arrs]).astype(np.float32)
    if it:
        iso.set_opt_state(flat(st.m), flat(st.v) if opt == "adam" else flat(st.m), np.array([st.bt1, st.bt2]))
    chi, ks = pkg.chis(iso), pkg.koopman(iso)
    ochi, oks = om(fx), O.expectation(om, fy)
    t = pkg.isotarget(iso)
    ot = O.isotarget(O.TransformISA(), om, fx, fy)
    A = O.myisa(oks.T)
    # the oracle's Float64 solve on the GPU's ks / chi: separates the forward's error from the device ISA chain's
    t_on_gpu_ks = O.fixperm(O.myisa(ks.astype(np.float64).T).T @ ks.astype(np.float64), chi.astype(np.float64))
    perm = np.arange(N)
    l = pkg.train_batch(iso, ot.astype(np.float32), perm=perm)
    p_before = om.get_flat().copy()
    lo = O.train_batch(om, fx, ot, st, 0, perm=perm)
    got, want = iso.model.get_params().astype(np.float64), om.get_flat()
    dstep = np.abs(want - p_before).max()
    print(f"it {it}: |chi err| {np.abs(chi - ochi).max():.2e} |ks err| {np.abs(ks - oks).max():.2e} kappa {np.abs(A).sum(axis=0).max():.0f} "
          f"|target err| {np.abs(t - ot).max():.2e} (chain alone {np.abs(t - t_on_gpu_ks).max():.2e}, scale {np.abs(ot).max():.1f}) "
          f"loss rel {abs(l - lo) / lo:.2e} |dtheta err| max {np.abs(got - want).max():.2e} (step {dstep:.2e}) "
          f"frac>1e-5: {np.mean(np.abs(got - want) > 1e-5):.2e}")
iso.close()
