"""CPU study: which stage's float32 rounding sets the "fp32 floor" of the C3 sub-sample runs (tests/golden/c3sub_reference.npz)?
The free-running trajectory of full_adam (ISA target, 10 full-batch iterations) is repeated with float32 rounding switched on
stage by stage, everything else in float64, and the end-of-run chi error against the all-float64 trajectory is printed.

    python scripts/floor_study.py [seeds]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import isokann_oracle as O  # noqa: E402
import benchshape_cases as bc  # noqa: E402

name = "full_adam"
niter, mb, opt, nout = bc.C3SUB_RUNS[name]
N, K = bc.C3SUB["N"], bc.C3SUB["K"]
fx, fy = bc.features64(bc.C3_ATOMS, bc.C3_F, N, K, bc.C3SUB["coords_seed"])
perm = np.arange(N)
r32 = lambda a: a.astype(np.float32).astype(np.float64)     # noqa: E731


def run(fwd="f64", out_round=False, noise=0.0, train="f64", theta32=False, seed=0, ks_round=False, target_round=False):
    """fwd: arithmetic of the forward that feeds the target ("f64" | "f32"); out_round: model outputs rounded to float32;
    noise: absolute gaussian noise added to every model output of the target stage; ks_round / target_round: K chi / the target
    rounded to float32; train: arithmetic of the training epoch; theta32: parameters and optimizer state stored in float32"""
    rng = np.random.default_rng(seed)
    om = bc.model(bc.c3sub_widths(name), bc.C3SUB["model_seed"], True)
    st = O.setup(O.AdamRegularized(), om)
    fx32, fy32 = fx.astype(np.float32), fy.astype(np.float32)
    for it in range(niter):
        if fwd == "f32":
            m = om.astype(np.float32)
            chi, outs = m(fx32).astype(np.float64), m(fy32.reshape(fy.shape[0], -1)).astype(np.float64)
        else:
            chi, outs = om(fx), om(fy.reshape(fy.shape[0], -1))
        if noise:
            chi = chi + noise * rng.normal(size=chi.shape)
            outs = outs + noise * rng.normal(size=outs.shape)
        if out_round:
            chi, outs = r32(chi), r32(outs)
        ks = outs.reshape(nout, K, N).mean(axis=1)
        if ks_round:
            ks = r32(ks)
        A = O.myisa(ks.T)
        t = O.fixperm(A.T @ ks, chi)
        if target_round:
            t = r32(t)
        if train == "f32":
            # the epoch in float32 arithmetic on float32 copies; the result goes back into the float64 containers
            m32 = om.astype(np.float32)
            s32 = O.OptState(st.rule, [a.astype(np.float32) for a in st.m], [a.astype(np.float32) for a in st.v], st.bt1, st.bt2)
            O.train_batch(m32, fx32, t.astype(np.float32), s32, 0, perm=perm)
            om.set_flat(m32.get_flat().astype(np.float64))
            st.m, st.v, st.bt1, st.bt2 = [a.astype(np.float64) for a in s32.m], [a.astype(np.float64) for a in s32.v], s32.bt1, s32.bt2
            continue
        O.train_batch(om, fx, t, st, 0, perm=perm)
        if theta32:
            om.set_flat(r32(om.get_flat()))
            st.m = [r32(a) for a in st.m]
            st.v = [r32(a) for a in st.v]
    return om(fx)


ref = run()
def report(label, **kw):
    errs = [np.abs(run(seed=s, **kw) - ref).max() for s in range(int(sys.argv[1]) if len(sys.argv) > 1 and kw.get("noise") else 1)]
    print(f"{label:70s} chi err after {niter} iterations: " + "  ".join(f"{e:.2e}" for e in errs), flush=True)

import os as _os
FULL = not _os.environ.get("FLOOR_STUDY_SHORT")
if FULL:
  report("outputs rounded to f32 (an ideal f32-output forward)", out_round=True)
  report("... + K chi rounded to f32", out_round=True, ks_round=True)
  report("... + target rounded to f32", out_round=True, ks_round=True, target_round=True)
  report("... + parameters and Adam state stored in f32", out_round=True, ks_round=True, target_round=True, theta32=True)
  report("only: parameters and Adam state stored in f32", theta32=True)
  report("only: target rounded to f32", target_round=True)
  report("only: K chi rounded to f32", ks_round=True)
  report("forward evaluated in f32 (numpy), rest f64", fwd="f32")
  report("training epoch in f32 (numpy), rest f64", train="f32")
  report("forward AND training in f32 (numpy)", fwd="f32", train="f32", ks_round=True, target_round=True)
  for nz in (1e-7, 2.5e-7, 5e-7):
      report(f"gaussian noise {nz:.1e} on every model output, outputs then rounded", noise=nz, out_round=True, ks_round=True)
print("--- K chi handed to the Float64 solve UNROUNDED (sum of the float32 outputs in double):")
for nz in (1e-7, 2.5e-7, 5e-7):
    report(f"gaussian noise {nz:.1e} on every model output, outputs rounded, K chi in double", noise=nz, out_round=True)
