"""Generates tests/golden/c3sub_reference.npz: fp64 oracle trajectories of run!() on the C3 sub-sample cases of
tests/benchshape_cases.py, and -- per case -- the FP32 FLOOR: how far the same oracle evaluated in float32 (NumPy/BLAS,
two different sample orders, i.e. two summation orders) ends up from its own fp64 run.  The GPU tests hold the HIP path to
max(north-star gate, 3 x floor): where the floor is above the gate (the ISA simplex at the plain initialisation amplifies
every rounding error of K chi by ~250) no fp32 evaluation can meet the gate, the reference's own CPU path included.

    python tests/golden/make_c3sub_reference.py          (about ten minutes of CPU)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import isokann_oracle as O  # noqa: E402
import benchshape_cases as bc  # noqa: E402


def run(name, dtype, order):
    niter, mb, opt, nout = bc.C3SUB_RUNS[name]
    N, K = bc.C3SUB["N"], bc.C3SUB["K"]
    fx, fy = bc.features64(bc.C3_ATOMS, bc.C3_F, N, K, bc.C3SUB["coords_seed"])
    om = bc.model(bc.c3sub_widths(name), bc.C3SUB["model_seed"], True).astype(dtype)
    rule = O.AdamRegularized() if opt == "adam" else O.NesterovRegularized()
    oiso = O.Iso(bc.oracle_data(fx[:, order], fy[:, :, order], dtype), om, O.setup(rule, om), minibatch=mb)
    inv = np.argsort(order)
    perms = [inv[p] for p in bc.run_perms(name)]            # the same samples per batch, whatever the storage order
    kappa = []
    for it in range(niter):
        ks = O.expectation(oiso.model, oiso.data.features[1]).astype(np.float64)
        # error amplification of the target transform: max column sum of the simplex inverse (ISA), 1 / range (shift-scale)
        kappa.append(np.abs(O.myisa(ks.T)).sum(axis=0).max() if nout > 1 else 1.0 / (ks.max() - ks.min()))
        O.run(oiso, 1, 1, perms=[perms[it]])
    return np.asarray(oiso.losses, np.float64), O.chis(oiso).astype(np.float64)[:, inv], np.asarray(kappa)


def main():
    out = {}
    N = bc.C3SUB["N"]
    for name in bc.C3SUB_RUNS:
        l64, c64, kappa = run(name, np.float64, np.arange(N))
        floor_l, floor_c = 0.0, 0.0
        for order in (np.arange(N), np.random.default_rng(1).permutation(N)):
            l32, c32, _ = run(name, np.float32, order)
            floor_l = max(floor_l, float(np.max(np.abs(l32 - l64) / np.abs(l64))))
            floor_c = max(floor_c, float(np.max(np.abs(c32 - c64))))
        p0 = bc.model(bc.c3sub_widths(name), bc.C3SUB["model_seed"], True).get_flat()
        out[name + "_losses"], out[name + "_chi"], out[name + "_kappa"] = l64, c64, kappa
        out[name + "_fp32_floor"] = np.array([floor_l, floor_c])
        out[name + "_p0_checksum"] = np.array([p0.sum(), np.abs(p0).sum()])
        print(f"{name}: losses {l64}\n   kappa {kappa}\n   fp32 floor: loss rel {floor_l:.2e}, chi {floor_c:.2e}", flush=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c3sub_reference.npz"), **out)


if __name__ == "__main__":
    main()
