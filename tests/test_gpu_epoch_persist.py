"""The persistent training-epoch kernel (kernels_train_epoch.hip: one launch per epoch, in-launch hand-offs between the
workgroups that own the layer-1 weight blocks and the ones that run the tails; option train_persist, off by default: it is
correct and measured slower than the launch chain) against the four-kernel launch chain and against the oracle's train_batch!
(src/iso.jl:179-194).

Tolerances: the two HIP paths differ only in summation order (K-slice partials, row split of the layer-1 gradient), so after a
few epochs their parameters agree to a few float32 ulps of the largest step; against the fp64 oracle the usual training
tolerances of test_gpu_parity.py apply."""
import os

import numpy as np
import pytest

from oracle import isokann_oracle as O

pytestmark = [pytest.mark.gpu, pytest.mark.own_fwd_mode, pytest.mark.usefixtures("fwd_mode", "experimental_lib")]


def _mk_model(pkg, om):
    m = pkg.densenet(om.widths, activation=om.activation, lastactivation=om.lastactivation, layernorm=om.layernorm)
    m.ln_mode = om.ln_mode
    m.params = om.get_flat().astype(np.float32)
    return m


def _rand_model(widths, seed, ln=True):
    rng = np.random.default_rng(seed)
    om = O.glorot_model(widths, rng, layernorm=ln)
    for b in om.b:
        b[...] = 0.1 * rng.normal(size=b.shape)
    if ln:
        om.ln_scale[...] = 1 + 0.1 * rng.normal(size=widths[0])
        om.ln_bias[...] = 0.1 * rng.normal(size=widths[0])
    return om


# ISK_TEST_PERSIST_MODE=2: the small-batch chain on the workgroups of one XCD (kernels_train_chain.hip; batches of at most 128 rows)
PMODE = int(os.environ.get("ISK_TEST_PERSIST_MODE", "1"))


CASES = [
    # atoms, F, N, K, widths tail, layernorm, minibatch, optimizer
    dict(A=20, F=60, N=130, K=3, widths=[60, 13, 4, 1], ln=True, mb=32, opt="adam"),          # ragged last unit (32 = 2 units), 4 steps
    dict(A=20, F=60, N=130, K=3, widths=[60, 13, 4, 1], ln=True, mb=50, opt="nesterov"),      # 50 rows: a half-filled unit
    dict(A=20, F=60, N=130, K=3, widths=[60, 13, 4, 1], ln=True, mb=0, opt="adam"),           # full batch: one step, 9 units
    dict(A=24, F=231, N=400, K=2, widths=[231, 38, 6, 1], ln=True, mb=100, opt="adam"),       # the ADP network (C2), 16 chunks x 3 tiles
    dict(A=24, F=200, N=300, K=2, widths=[200, 142, 12, 3], ln=True, mb=100, opt="adam"),     # villin's tail: 9 tiles, three outputs
    dict(A=24, F=200, N=300, K=2, widths=[200, 40, 3], ln=False, mb=64, opt="nesterov"),      # no LayerNorm, one tail layer
    dict(A=24, F=100, N=256, K=2, widths=[100, 20, 10, 5, 2], ln=True, mb=64, opt="adam"),    # three tail layers
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "F%d-%s-mb%d-%s%s" % (c["F"], "x".join(map(str, c["widths"][1:])), c["mb"], c["opt"], "" if c["ln"] else "-noln"))
def test_persistent_epoch_matches_launch_chain_and_oracle(gpu_pkg, case):
    pkg = gpu_pkg
    c = case
    if PMODE == 2 and (c["mb"] == 0 or c["mb"] > 128):
        pytest.skip("the one-XCD chain covers batches of at most 128 rows")
    xs, ys = O.synthetic_coords(c["A"], c["N"], c["K"], seed=3)
    pairs = O.random_pairs(c["A"], c["F"], seed=4)
    om = _rand_model(c["widths"], seed=5, ln=c["ln"])
    nout = c["widths"][-1]
    feat = O.FeaturesPairs(pairs)
    odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64), feat)
    rule = O.OptRule(c["opt"], 1e-2, 1e-3)
    st = O.setup(rule, om)
    isos = {}
    for persist in (PMODE, 0):
        isos[1 if persist else 0] = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_mk_model(pkg, om), opt=pkg.OptSpec(c["opt"], 1e-2, 1e-3),
                                minibatch=c["mb"], options={"train_persist": persist})
    rng = np.random.default_rng(0)
    for ep in range(3):
        tgt = np.asarray(rng.uniform(size=(nout, c["N"])), np.float32)
        perm = rng.permutation(c["N"])
        lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), st, c["mb"], perm=perm)
        ls = {}
        for persist, iso in isos.items():
            ls[persist] = pkg.train_batch(iso, tgt, perm=perm)
            assert iso.get_option("train_persist_active") == (1 if persist else 0), "the epoch did not take the path under test"
        assert abs(ls[1] - lo) <= 1e-4 * abs(lo), (ep, ls, lo)
        assert abs(ls[1] - ls[0]) <= 2e-6 * abs(lo), (ep, ls)
    p1, p0 = isos[1].model.get_params(), isos[0].model.get_params()
    ref = om.get_flat()
    np.testing.assert_allclose(p1, ref, atol=3e-5, rtol=3e-4)
    if c["opt"] == "nesterov":
        np.testing.assert_allclose(p1, p0, atol=2e-6, rtol=2e-5)
    else:
        # Adam divides by sqrt(v): where the second moment is at rounding level a step's direction is not determined
        bad = np.abs(p1 - p0) > 2e-6 + 2e-5 * np.abs(p0)
        assert bad.mean() < 0.02, bad.mean()
    m1, v1, bt1 = isos[1].get_opt_state()
    m0, v0, bt0 = isos[0].get_opt_state()
    assert bt1[0] == pytest.approx(bt0[0], rel=1e-12) and bt1[1] == pytest.approx(bt0[1], rel=1e-12)
    np.testing.assert_allclose(m1, m0, atol=2e-6, rtol=1e-3)
    for iso in isos.values():
        iso.close()


def test_persistent_epoch_nonfinite_loss_keeps_weights(gpu_pkg):
    pkg = gpu_pkg
    xs, ys = O.synthetic_coords(10, 40, 2, seed=3)
    pairs = O.random_pairs(10, 20, seed=4)
    om = _rand_model([20, 8, 3, 1], seed=2)
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_mk_model(pkg, om), minibatch=16, options={"train_persist": PMODE})
    before = iso.model.get_params()
    tgt = np.full((1, 40), np.nan, np.float32)
    with pytest.raises(pkg.DomainError) as ei:
        pkg.train_batch(iso, tgt, perm=np.arange(40))
    assert iso.get_option("train_persist_active") == 1
    assert ei.value.status == pkg.capi.ECOLLAPSED
    np.testing.assert_array_equal(iso.model.get_params(), before)
    iso.close()


def test_persistent_epoch_in_run_with_isa_target(gpu_pkg):
    """run! with the ISA target and per-iteration featurization (the bench's schedule: the side-stream featurizer shares the CUs
    with the persistent launch) against the oracle"""
    pkg = gpu_pkg
    A, F, N, K, nout = 24, 120, 384, 4, 3
    xs, ys = O.synthetic_coords(A, N, K, seed=7)
    pairs = O.random_pairs(A, F, seed=8)
    widths = O.pairnet_widths(F, nout=nout)
    om = _rand_model(widths, seed=9)
    odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64), O.FeaturesPairs(pairs))
    oiso = O.Iso(odata, om, O.setup(O.AdamRegularized(), om), minibatch=64, target=O.TransformISA())
    rng = np.random.default_rng(1)
    perms = np.stack([rng.permutation(N) for _ in range(3)]).astype(np.int32)
    m0 = _mk_model(pkg, om)
    O.run(oiso, 3, 1, perms=[p.astype(np.int64) for p in perms])
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=m0, opt=pkg.AdamRegularized(), minibatch=64,
                  target=pkg.TransformISA(), featurize_every_iteration=True, options={"train_persist": PMODE})
    pkg.run(iso, 3, 1, perms=perms)
    assert iso.get_option("train_persist_active") == 1
    for a, b in zip(iso.losses, oiso.losses):
        assert abs(a - b) <= 2e-3 * abs(b), (iso.losses, oiso.losses)
    np.testing.assert_allclose(pkg.chis(iso), O.chis(oiso), atol=2e-4)
    iso.close()
