"""Generates tests/golden/c3sub_reference.npz: fp64 oracle trajectories of run!() on the C3 sub-sample cases of
tests/benchshape_cases.py, and -- per case -- the FP32 FLOOR: how far the same oracle evaluated in float32 (NumPy/BLAS,
eight summation orders: sample order x feature order) ends up from its own fp64 run -- a distribution (every sample is stored), whose
largest member is the floor.  The GPU tests hold the HIP path, in every forward arithmetic, to max(north-star gate, 1 x floor): where the floor is above the gate (the ISA simplex at the plain initialisation amplifies
every rounding error of K chi by ~250) no fp32 evaluation can meet the gate, the reference's own CPU path included.

    python tests/golden/make_c3sub_reference.py          (about forty minutes of CPU)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import isokann_oracle as O  # noqa: E402
import benchshape_cases as bc  # noqa: E402


def run(name, dtype, order, forder=None):
    """order: storage order of the samples (changes the summation order of every sum over samples: gradients, K-means' neighbours in
    BLAS blocks); forder: storage order of the FEATURES -- the same network on permuted inputs (first-layer columns and LayerNorm
    affine permuted with them): changes the summation order of the layer-1 dot products and of the LayerNorm statistics"""
    niter, mb, opt, nout = bc.C3SUB_RUNS[name]
    N, K = bc.C3SUB["N"], bc.C3SUB["K"]
    fx, fy = bc.features64(bc.C3_ATOMS, bc.C3_F, N, K, bc.C3SUB["coords_seed"])
    om = bc.model(bc.c3sub_widths(name), bc.C3SUB["model_seed"], True)
    if forder is not None:
        fx, fy = fx[forder], fy[forder]
        om.W[0] = np.ascontiguousarray(om.W[0][:, forder])
        om.ln_scale, om.ln_bias = om.ln_scale[forder].copy(), om.ln_bias[forder].copy()
    om = om.astype(dtype)
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


NORD = 8


def fx_rows():
    return bc.C3_F


def main():
    out = {}
    N = bc.C3SUB["N"]
    for name in bc.C3SUB_RUNS:
        l64, c64, kappa = run(name, np.float64, np.arange(N))
        # The floor is a DISTRIBUTION: the trajectory amplifies rounding-level differences of K chi a few hundredfold
        # (scripts/floor_study.py: the Float32 rounding of K chi alone, everything else exact, ends 2.5e-5 away after 10 full-batch
        # iterations), so two float32 evaluations that differ only in summation order end up to ~4x apart.  NORD evaluations
        # (sample order x feature order); the bound of the GPU tests is the largest, every sample is kept in the file.
        F = fx_rows()
        samples = []
        for k in range(NORD):
            order = np.arange(N) if k % 2 == 0 else np.random.default_rng(1 + k).permutation(N)
            forder = None if k < 2 else np.random.default_rng(50 + k).permutation(F)
            l32, c32, _ = run(name, np.float32, order, forder)
            samples.append([float(np.max(np.abs(l32 - l64) / np.abs(l64))), float(np.max(np.abs(c32 - c64)))])
            print(f"   {name} float32 evaluation {k}: loss rel {samples[-1][0]:.2e}, chi {samples[-1][1]:.2e}", flush=True)
        floor_l, floor_c = (float(v) for v in np.max(np.asarray(samples), axis=0))
        out[name + "_fp32_floor_samples"] = np.asarray(samples)
        p0 = bc.model(bc.c3sub_widths(name), bc.C3SUB["model_seed"], True).get_flat()
        out[name + "_losses"], out[name + "_chi"], out[name + "_kappa"] = l64, c64, kappa
        out[name + "_fp32_floor"] = np.array([floor_l, floor_c])
        out[name + "_p0_checksum"] = np.array([p0.sum(), np.abs(p0).sum()])
        print(f"{name}: losses {l64}\n   kappa {kappa}\n   fp32 floor: loss rel {floor_l:.2e}, chi {floor_c:.2e}", flush=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c3sub_reference.npz"), **out)


if __name__ == "__main__":
    main()
