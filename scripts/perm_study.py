"""GPU diagnostic: is the end-of-run chi error of a C3 sub-sample run (tests/golden/c3sub_reference.npz) a property of the forward
ARITHMETIC, or of the summation order?  The same network on PERMUTED features (pair list, first-layer columns and LayerNorm affine
permuted together: the same function, another order of every layer-1 dot product and LayerNorm sum) in every forward mode.
    python scripts/perm_study.py [case] [ntrials]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import benchshape_cases as bc  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "full_adam"
ntr = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pkg = ge.load_package()
gold = np.load(os.path.join(ROOT, "tests", "golden", "c3sub_reference.npz"))
niter, mb, opt, nout = bc.C3SUB_RUNS[name]
N, K = bc.C3SUB["N"], bc.C3SUB["K"]
xs, ys, pairs = bc.coords(bc.C3_ATOMS, bc.C3_F, N, K, bc.C3SUB["coords_seed"])
pairs = np.asarray(pairs)
print(f"{name}: CPU float32 floor samples (chi): " + " ".join(f"{v:.2e}" for v in gold[name + "_fp32_floor_samples"][:, 1])
      if name + "_fp32_floor_samples" in gold else f"floor {gold[name + '_fp32_floor'][1]:.2e}")
for mode in (0, 1, 3):
    errs = []
    for t in range(ntr):
        om = bc.model(bc.c3sub_widths(name), bc.C3SUB["model_seed"], True)
        fo = np.arange(bc.C3_F) if t == 0 else np.random.default_rng(900 + t).permutation(bc.C3_F)
        om.W[0] = np.ascontiguousarray(om.W[0][:, fo])
        om.ln_scale, om.ln_bias = om.ln_scale[fo].copy(), om.ln_bias[fo].copy()
        m = pkg.densenet(om.widths, layernorm=True)
        m.params = om.get_flat().astype(np.float32)
        popt = pkg.AdamRegularized() if opt == "adam" else pkg.NesterovRegularized()
        iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs([tuple(p) for p in pairs[fo]])), model=m, opt=popt, minibatch=mb, forward_mode=mode)
        pkg.run(iso, niter, 1, perms=bc.run_perms(name))
        errs.append(float(np.max(np.abs(pkg.chis(iso) - gold[name + "_chi"]))))
        iso.close()
    print(f"forward mode {mode}: chi error after {niter} iterations, feature order 0 (as stored) .. {ntr - 1}: " + "  ".join(f"{e:.2e}" for e in errs), flush=True)
