"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU
oracle on the same seeded inputs, and against the committed golden vectors.

Tolerances (fp32 path vs the fp64 oracle), per BASELINE.json north_star / SURVEY 8d:
  features      rel 2e-6       (direct-difference pdist, one sqrt)
  model output  abs 2e-5       (K <= 1712 f32 MFMA fma-chain + sigmoid)
  chi after training   abs 1e-5 at minibatch = full batch / small nets (north-star gate),
                       abs 5e-5 for the multi-iteration golden runs
  losses        rel 1e-4
"""
import os
import sys

import numpy as np
import pytest

from oracle import isokann_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fwd_mode")]

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _mk_model(pkg, om: O.Model):
    m = pkg.densenet(om.widths, activation=om.activation, lastactivation=om.lastactivation, layernorm=om.layernorm)
    m.ln_mode = om.ln_mode
    m.params = om.get_flat().astype(np.float32)
    return m


def _rand_model(widths, seed, ln=True, act="sigmoid", lastact="identity"):
    rng = np.random.default_rng(seed)
    om = O.glorot_model(widths, rng, layernorm=ln, activation=act, lastactivation=lastact)
    for b in om.b:
        b[...] = 0.1 * rng.normal(size=b.shape)
    if ln:
        om.ln_scale[...] = 1 + 0.1 * rng.normal(size=widths[0])
        om.ln_bias[...] = 0.1 * rng.normal(size=widths[0])
    return om


# ------------------------------------------------------------------ featurizer --------
@pytest.mark.parametrize("A,N,K", [(4, 5, 3), (22, 100, 4), (57, 33, 2)])
def test_features_allpairs(gpu_pkg, A, N, K):
    pkg = gpu_pkg
    xs, ys = O.synthetic_coords(A, N, K, seed=A)
    data = pkg.SimulationData(xs, ys, pkg.FeaturesAll())
    iso = pkg.Iso(data, minibatch=0)
    f = pkg.features(iso)
    want = O.flatpairdists(np.asarray(xs, np.float64))
    assert f.shape == want.shape == (A * (A - 1) // 2, N)
    np.testing.assert_allclose(f, want, rtol=2e-6, atol=1e-7)
    fy = pkg.propfeatures(iso)
    np.testing.assert_allclose(fy, O.flatpairdists(np.asarray(ys, np.float64)), rtol=2e-6, atol=1e-7)
    iso.close()


def test_features_pairlist_and_atoms(gpu_pkg):
    pkg = gpu_pkg
    A, N, K = 40, 64, 3
    xs, ys = O.synthetic_coords(A, N, K, seed=9)
    pairs = O.random_pairs(A, 100, seed=3) + [(7, 7), (5, 2)]      # a zero distance and an unsorted pair
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), minibatch=0)
    np.testing.assert_allclose(pkg.features(iso), O.pdists(np.asarray(xs, np.float64), pairs), rtol=2e-6, atol=1e-7)
    iso.close()
    atoms = [3, 0, 17, 21, 39]
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesAtoms(atoms)), minibatch=0)
    np.testing.assert_allclose(pkg.features(iso), O.flatpairdists(np.asarray(xs, np.float64), atoms), rtol=2e-6, atol=1e-7)
    iso.close()


@pytest.mark.gpu
@pytest.mark.parametrize("A,F,N,K", [(600, 1700, 3, 1), (683, 2048, 5, 2), (9, 33, 1030, 1)])
def test_features_pairlist_edge_shapes(gpu_pkg, A, F, N, K):
    """rows go through buffer instructions with a per-row descriptor: fewer samples than waves, a feature count that is /
    is not a multiple of the 64-lane store width, the largest register-resident atom count, more rows than one trip"""
    pkg = gpu_pkg
    xs, ys = O.synthetic_coords(A, N, K, seed=A)
    pairs = O.random_pairs(A, F, seed=5)
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), minibatch=0)
    np.testing.assert_allclose(pkg.features(iso), O.pdists(np.asarray(xs, np.float64), pairs), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(pkg.propfeatures(iso), O.pdists(np.asarray(ys, np.float64), pairs), rtol=2e-6, atol=1e-7)
    iso.close()


# ------------------------------------------------------------------ forward -----------
@pytest.mark.parametrize("widths,ln,act", [
    ([2, 2, 1, 1], True, "sigmoid"),            # C1 default net
    ([231, 38, 6, 1], True, "sigmoid"),         # C2 default net
    ([100, 22, 5, 3], True, "sigmoid"),
    ([33, 160, 17, 40, 2], True, "tanh"),       # widest supported layer 1, deeper tail
    ([48, 16, 1], False, "relu"),
    ([20, 7], True, "sigmoid"),                 # single Dense layer
    ([65, 30, 9, 2], True, "swish"),
])
@pytest.mark.parametrize("mode", ["var_eps2", "std_eps"])
def test_model_forward(gpu_pkg, widths, ln, act, mode):
    pkg = gpu_pkg
    if not ln and mode != "var_eps2":
        pytest.skip("mode irrelevant without LayerNorm")
    om = _rand_model(widths, seed=sum(widths), ln=ln, act=act)
    om.ln_mode = mode
    rng = np.random.default_rng(1)
    N, K = 70, 3
    xs = rng.normal(size=(widths[0], N)) + 1.0
    ys = rng.normal(size=(widths[0], K, N)) + 1.0
    nout = widths[-1]
    tgt = pkg.TransformShiftscale() if nout == 1 else pkg.TransformISA()
    iso = pkg.Iso(pkg.SimulationData(xs, ys), model=_mk_model(pkg, om), target=tgt, minibatch=0)
    x32 = np.asarray(xs, np.float32).astype(np.float64)
    y32 = np.asarray(ys, np.float32).astype(np.float64)
    np.testing.assert_allclose(iso.model(xs), om(x32), atol=2e-5, rtol=1e-5)
    out3 = iso.model(ys)
    assert out3.shape == (nout, K, N)
    np.testing.assert_allclose(out3, om(y32), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(pkg.chis(iso), om(x32), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(pkg.koopman(iso), O.expectation(om, y32), atol=2e-5, rtol=1e-5)
    iso.close()


def test_forward_ragged_row_counts(gpu_pkg):
    """every row count around the 16-row MFMA unit and the 2-unit pass"""
    pkg = gpu_pkg
    om = _rand_model([37, 20, 4, 1], seed=2)
    rng = np.random.default_rng(3)
    xs = rng.normal(size=(37, 4)) + 1
    ys = rng.normal(size=(37, 2, 4)) + 1
    iso = pkg.Iso(pkg.SimulationData(xs, ys), model=_mk_model(pkg, om), minibatch=0)
    for n in [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, 65, 1000, 4097]:
        x = np.asarray(rng.normal(size=(37, n)) + 1, np.float32)
        np.testing.assert_allclose(iso.model(x), om(x.astype(np.float64)), atol=2e-5, rtol=1e-5)
    iso.close()


# ------------------------------------------------------------------ targets -----------
def _coords_case(pkg, A, F, N, K, nout, seed, layers=3):
    xs, ys = O.synthetic_coords(A, N, K, seed=seed)
    pairs = O.random_pairs(A, F, seed=seed + 1)
    widths = O.pairnet_widths(F, layers=layers, nout=nout)
    om = _rand_model(widths, seed=seed + 2)
    feat = O.FeaturesPairs(pairs)
    odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64), feat)
    data = pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs))
    return data, odata, om


def test_isotarget_shiftscale(gpu_pkg):
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 30, 120, 200, 5, 1, seed=4)
    iso = pkg.Iso(data, model=_mk_model(pkg, om), minibatch=0)
    t = pkg.isotarget(iso)
    want = O.isotarget(O.TransformShiftscale(), om, *odata.features)
    assert t.min() == 0.0 and t.max() == 1.0
    np.testing.assert_allclose(t, want, atol=2e-5)
    iso.close()


@pytest.mark.parametrize("nout,whitening,permute", [(2, False, True), (3, False, True), (3, True, True), (4, False, False)])
def test_isotarget_isa(gpu_pkg, nout, whitening, permute):
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 30, 150, 300, 4, nout, seed=5 + nout)
    tr = pkg.TransformISA(permute=permute, whitening=whitening)
    iso = pkg.Iso(data, model=_mk_model(pkg, om), target=tr, minibatch=0)
    t = pkg.isotarget(iso)
    want = O.isotarget(O.TransformISA(permute=permute, whitening=whitening), om, *odata.features)
    # the ISA map amplifies fp32 forward error by |inv(X[i,:])|; compare relative to the target scale
    scale = np.abs(want).max()
    np.testing.assert_allclose(t, want, atol=2e-4 * scale)
    # known answer: the selected columns form the identity (up to the row permutation)
    ks = pkg.koopman(iso).astype(np.float64)
    X = ks.T
    if whitening:                       # rows are chosen on the whitened coordinates (src/isotarget.jl:85-89)
        ev, U = np.linalg.eigh(X.T @ X / X.shape[0])
        X = X @ ((U * ev ** -0.5) @ U.T)
    idx = O.indexmap(X)
    sel = t[:, idx]
    assert np.allclose(np.sort(sel, axis=0)[-1], 1.0, atol=1e-4) and np.allclose(np.abs(sel).sum(axis=0), 1.0, atol=1e-3)
    iso.close()


def test_constant_chi_raises_domain_error(gpu_pkg):
    pkg = gpu_pkg
    om = _rand_model([6, 3, 1], seed=1)
    for w in om.W:
        w[...] = 0
    xs = np.random.default_rng(0).normal(size=(6, 20))
    ys = np.random.default_rng(1).normal(size=(6, 3, 20))
    iso = pkg.Iso(pkg.SimulationData(xs, ys), model=_mk_model(pkg, om), minibatch=0)
    with pytest.raises(pkg.DomainError) as ei:
        pkg.isotarget(iso)
    assert ei.value.status == pkg.capi.ECONSTANT_CHI
    with pytest.raises(pkg.DomainError):
        pkg.run(iso, 1)
    iso.close()


def test_singular_isa_raises_domain_error(gpu_pkg):
    pkg = gpu_pkg
    om = _rand_model([6, 3, 2], seed=1)
    om.W[-1][...] = 0        # both outputs constant -> collapsed subspace
    xs = np.random.default_rng(0).normal(size=(6, 20))
    ys = np.random.default_rng(1).normal(size=(6, 3, 20))
    iso = pkg.Iso(pkg.SimulationData(xs, ys), model=_mk_model(pkg, om), minibatch=0)
    with pytest.raises(pkg.DomainError) as ei:
        pkg.isotarget(iso)
    assert ei.value.status == pkg.capi.ESINGULAR
    iso.close()


@pytest.mark.parametrize("nout,N", [(2, 20), (3, 2500), (4, 1025)])
def test_nan_model_isa_raises_domain_error(gpu_pkg, nout, N):
    """a diverged model (NaN parameter: every Koopman value NaN) must come back as the reference's DomainError
    (src/isotarget.jl:94-97), not as an out-of-bounds access: no row is an argmax candidate in any of the d rounds,
    with one and with several row slabs; run() reports it too, and the weights stay as they were"""
    pkg = gpu_pkg
    om = _rand_model([6, 5, nout], seed=nout)
    om.b[0][1] = np.nan
    rng = np.random.default_rng(0)
    xs, ys = rng.normal(size=(6, N)), rng.normal(size=(6, 2, N))
    iso = pkg.Iso(pkg.SimulationData(xs, ys), model=_mk_model(pkg, om), minibatch=0)
    with pytest.raises(pkg.DomainError) as ei:
        pkg.isotarget(iso)
    assert ei.value.status == pkg.capi.ESINGULAR
    before = iso.model.get_params()
    with pytest.raises(pkg.DomainError):
        pkg.run(iso, 1, 1)
    np.testing.assert_array_equal(np.isnan(iso.model.get_params()), np.isnan(before))
    iso.close()


# ------------------------------------------------------------------ gradients ---------
@pytest.mark.parametrize("widths,ln,act,B", [
    ([50, 14, 5, 3], True, "sigmoid", 40),
    ([50, 14, 5, 3], False, "sigmoid", 17),
    ([231, 38, 6, 1], True, "sigmoid", 100),
    ([33, 100, 20, 9, 2], True, "tanh", 64),
    ([20, 4], True, "sigmoid", 16),
    ([24, 9, 2], True, "swish", 33),
    ([64, 160, 12, 3], True, "sigmoid", 48),            # widest fused first layer: two groups of five output tiles in train_fwd1n
    ([40, 24, 50, 30, 20, 40], True, "tanh", 37),       # four tail layers (two beyond the prefetched ones), 40 outputs (> 32: plain loss loop)
    ([300, 70, 8, 2], False, "relu", 128),              # 19 feature chunks: K slices of unequal length, no LayerNorm
])
def test_batch_gradients(gpu_pkg, widths, ln, act, B):
    pkg = gpu_pkg
    om = _rand_model(widths, seed=B, ln=ln, act=act)
    rng = np.random.default_rng(B)
    N = 128
    xs = rng.normal(size=(widths[0], N)) + 1.0
    ys = rng.normal(size=(widths[0], 2, N)) + 1.0
    nout = widths[-1]
    tgt = np.asarray(rng.uniform(size=(nout, N)), np.float32)
    tr = pkg.TransformShiftscale() if nout == 1 else (pkg.TransformISA() if nout <= 8 else pkg.TransformPseudoInv())
    iso = pkg.Iso(pkg.SimulationData(xs, ys), model=_mk_model(pkg, om), target=tr, minibatch=0)
    idx = rng.permutation(N)[:B]
    g, l = pkg.batch_grads(iso, tgt, idx)
    x32 = np.asarray(xs, np.float32).astype(np.float64)
    t64 = tgt.astype(np.float64)
    w = O.loss_weights(t64) if nout > 1 else np.ones(1)
    lo, go, _ = O.batch_loss_and_grads(om, x32[:, idx], t64[:, idx], w)
    gflat = np.concatenate([a.reshape(-1, order="F") for a in go])
    assert abs(l - lo) <= 2e-5 * abs(lo)
    scale = np.abs(gflat).max()
    np.testing.assert_allclose(g, gflat, atol=3e-5 * scale, rtol=1e-4)
    iso.close()


# ------------------------------------------------------------------ training ----------
@pytest.mark.parametrize("opt", ["adam", "nesterov"])
@pytest.mark.parametrize("minibatch", [0, 32, 50])
def test_train_batch_epoch(gpu_pkg, opt, minibatch):
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 20, 60, 130, 3, 1, seed=11)
    rule = O.OptRule(opt, 1e-2, 1e-3)
    popt = pkg.OptSpec(opt, 1e-2, 1e-3)
    iso = pkg.Iso(data, model=_mk_model(pkg, om), opt=popt, minibatch=minibatch)
    rng = np.random.default_rng(0)
    st = O.setup(rule, om)
    for ep in range(3):
        tgt = np.asarray(rng.uniform(size=(1, 130)), np.float32)
        perm = rng.permutation(130)
        l = pkg.train_batch(iso, tgt, perm=perm)
        lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), st, minibatch, perm=perm)
        assert abs(l - lo) <= 1e-4 * abs(lo)
    np.testing.assert_allclose(iso.model.get_params(), om.get_flat(), atol=2e-5, rtol=2e-4)
    m, v, bt = iso.get_opt_state()
    np.testing.assert_allclose(m, np.concatenate([a.reshape(-1, order="F") for a in st.m]), atol=1e-5, rtol=1e-3)
    if opt == "adam":
        assert bt[0] == pytest.approx(st.bt1) and bt[1] == pytest.approx(st.bt2)
    iso.close()


@pytest.mark.gpu
def test_train_batch_changing_batch_plan_and_restored_optimizer_state(gpu_pkg):
    """the training epoch is replayed as a captured graph: a different batch plan re-captures it, and an optimizer state
    written back from the host (Adam step products included) must reach the device copy the graph reads"""
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 20, 60, 130, 3, 1, seed=13)
    rule = O.OptRule("adam", 1e-2, 1e-3)
    iso = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.OptSpec("adam", 1e-2, 1e-3), minibatch=32)
    rng = np.random.default_rng(1)
    st = O.setup(rule, om)
    for k, mb in enumerate([32, 32, 50, 32, 0, 50, 50]):
        tgt = np.asarray(rng.uniform(size=(1, 130)), np.float32)
        perm = rng.permutation(130)
        l = pkg.train_batch(iso, tgt, minibatch=mb, perm=perm)
        lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), st, mb, perm=perm)
        assert abs(l - lo) <= 1e-4 * abs(lo), (k, mb)
        if k == 3:                                  # round trip of the optimizer state through the host
            m, v, bt = iso.get_opt_state()
            iso.set_opt_state(m, v, bt)
    np.testing.assert_allclose(iso.model.get_params(), om.get_flat(), atol=3e-5, rtol=3e-4)
    _, _, bt = iso.get_opt_state()
    assert bt[0] == pytest.approx(st.bt1) and bt[1] == pytest.approx(st.bt2)
    iso.close()


def test_nonfinite_loss_raises_and_keeps_weights(gpu_pkg):
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 10, 20, 40, 2, 1, seed=3)
    iso = pkg.Iso(data, model=_mk_model(pkg, om), minibatch=0)
    before = iso.model.get_params()
    tgt = np.full((1, 40), np.nan, np.float32)
    with pytest.raises(pkg.DomainError) as ei:
        pkg.train_batch(iso, tgt, perm=np.arange(40))
    assert ei.value.status == pkg.capi.ECOLLAPSED
    np.testing.assert_array_equal(iso.model.get_params(), before)
    iso.close()


def test_run_full_batch_chi_within_1e5(gpu_pkg):
    """north-star gate: chi within 1e-5 of the CPU oracle after 10 iterations at minibatch=0"""
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 22, 231, 256, 5, 1, seed=21)
    iso = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(), minibatch=0)
    oiso = O.Iso(odata, om, O.setup(O.AdamRegularized(), om), minibatch=0)
    perms = np.stack([np.arange(256)] * 10)
    pkg.run(iso, 10, 1, perms=perms)
    O.run(oiso, 10, 1, perms=list(perms))
    assert np.max(np.abs(pkg.chis(iso) - O.chis(oiso))) <= 1e-5
    np.testing.assert_allclose(iso.losses, oiso.losses, rtol=1e-4)
    iso.close()


@pytest.mark.parametrize("every_iter", [False, True])
@pytest.mark.parametrize("N,K,nout", [(37, 3, 1), (21, 5, 3), (17, 1, 1), (33, 7, 3)])
def test_run_when_ys_and_xs_rows_share_a_plane_unit(gpu_pkg, N, K, nout, every_iter):
    """the split-f16 operand planes are stored in units of 16 rows over the [ys; xs] slab (isk_plane_off): N*K odd or not a
    multiple of 16 puts the xs rows in the middle of a unit and the featurizer's row pairs across the ys / xs boundary;
    the side-stream launch of the per-iteration mode covers both parts in one go, the resident one in two"""
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 13, 45, N, K, nout, seed=300 + N)
    tgt, otgt = (pkg.TransformShiftscale(), O.TransformShiftscale()) if nout == 1 else (pkg.TransformISA(), O.TransformISA())
    iso = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-2, 1e-3), target=tgt, minibatch=8,
                  featurize_every_iteration=every_iter)
    oiso = O.Iso(odata, om, O.setup(O.AdamRegularized(1e-2, 1e-3), om), target=otgt, minibatch=8)
    np.testing.assert_allclose(pkg.chis(iso), O.chis(oiso), atol=2e-5)
    np.testing.assert_allclose(pkg.koopman(iso), O.expectation(om, odata.features[1]), atol=2e-5)
    rng = np.random.default_rng(N)
    perms = np.stack([rng.permutation(N) for _ in range(6)])
    pkg.run(iso, 6, 1, perms=perms)                     # >= 4 optimizer steps per iteration: the side-stream featurizer is used
    O.run(oiso, 6, 1, perms=list(perms))
    np.testing.assert_allclose(iso.losses, oiso.losses, rtol=2e-3)
    np.testing.assert_allclose(pkg.chis(iso), O.chis(oiso), atol=2e-4)
    iso.close()


@pytest.mark.parametrize("every_iter", [False, True])
@pytest.mark.parametrize("name", ["c1_shiftscale_nesterov", "c2small_shiftscale_adam", "c3small_isa_adam"])
def test_run_matches_golden(gpu_pkg, name, every_iter):
    pkg = gpu_pkg
    g = np.load(os.path.join(GOLD, name + ".npz"))
    kind = int(g["featurizer_kind"])
    feat = {0: pkg.identity, 1: pkg.FeaturesAll(), 2: pkg.FeaturesPairs([tuple(p) for p in g["pairs"]])}[kind]
    widths = [int(w) for w in g["widths"]]
    model = pkg.densenet(widths, layernorm=True)
    model.params = g["params0"].astype(np.float32)
    opt = pkg.OptSpec(str(g["opt_kind"]), float(g["opt_lr"]), float(g["opt_reg"]))
    iso = pkg.Iso(pkg.SimulationData(g["xs"], g["ys"], feat), model=model, opt=opt, minibatch=int(g["minibatch"]),
                  featurize_every_iteration=every_iter)
    t0 = pkg.isotarget(iso)
    scale = max(1.0, np.abs(g["target0"]).max())
    np.testing.assert_allclose(t0, g["target0"], atol=2e-4 * scale)
    pkg.run(iso, int(g["n_iter"]), 1, perms=g["perms"])
    np.testing.assert_allclose(iso.losses, g["losses"], rtol=2e-4)
    np.testing.assert_allclose(pkg.chis(iso), g["chis"], atol=5e-5)
    np.testing.assert_allclose(pkg.koopman(iso), g["koopman"], atol=5e-5)
    iso.close()


def test_engine_shuffle_is_seeded_and_valid(gpu_pkg):
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 10, 30, 90, 2, 1, seed=8)
    a = pkg.Iso(data, model=_mk_model(pkg, om), minibatch=16, seed=7)
    b = pkg.Iso(data, model=_mk_model(pkg, om), minibatch=16, seed=7)
    c = pkg.Iso(data, model=_mk_model(pkg, om), minibatch=16, seed=8)
    for i in (a, b, c):
        pkg.run(i, 3)
    assert a.losses == b.losses and a.losses != c.losses
    assert all(np.isfinite(a.losses))
    for i in (a, b, c):
        i.close()


def test_argument_errors(gpu_pkg):
    pkg = gpu_pkg
    xs = np.zeros((6, 10), np.float32)
    ys = np.zeros((6, 2, 10), np.float32)
    with pytest.raises(pkg.IskError):        # shiftscale needs nout == 1
        pkg.Iso(pkg.SimulationData(xs, ys), model=pkg.densenet([6, 3, 2]), target=pkg.TransformShiftscale())
    with pytest.raises(pkg.IskError):        # ISA needs nout > 1
        pkg.Iso(pkg.SimulationData(xs, ys), model=pkg.densenet([6, 3, 1]), target=pkg.TransformISA())
    iso = pkg.Iso(pkg.SimulationData(xs + 1, ys + 1), model=pkg.densenet([6, 3, 1]))
    with pytest.raises(pkg.IskError):
        iso.model(np.zeros((5, 3), np.float32))
    with pytest.raises(pkg.IskError):        # no target yet
        pkg.train_batch(iso, None)
    iso.close()


@pytest.mark.parametrize("nout,minibatch", [(1, 32), (3, 0), (3, 48)])
def test_dp_driver_single_rank_matches_run(gpu_pkg, nout, minibatch):
    """the data-parallel driver (dp.py + isk_dp_* entry points) with one rank and no process group
    must reproduce isk_run on the same injected shuffles"""
    import importlib
    import torch
    pkg = gpu_pkg
    dp = importlib.import_module("isokann_jl_amd.dp")
    data, odata, om = _coords_case(pkg, 16, 40, 144, 3, nout, seed=30 + nout)
    rng = np.random.default_rng(5)
    perms = [rng.permutation(144) for _ in range(3)]
    a = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-2, 1e-3), minibatch=minibatch)
    b = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-2, 1e-3), minibatch=minibatch)
    pkg.run(a, 3, 1, perms=np.stack(perms))
    torch.cuda.set_device(0)
    local = dp.HipLocal(b)
    losses = dp.dp_run(local, dp.Comm(device=torch.device("cuda", 0)), 3, 1, minibatch, rng, perms=perms)
    np.testing.assert_allclose(losses, a.losses, rtol=2e-5)
    # ISA on the host (gathered path) vs the device kernel: same algorithm, Float64 both
    np.testing.assert_allclose(b.model.get_params(), a.model.get_params(), atol=2e-6, rtol=2e-4)
    np.testing.assert_allclose(pkg.chis(b), pkg.chis(a), atol=1e-5)
    a.close()
    b.close()


@pytest.mark.parametrize("nout", [1, 3])
def test_dp_native_rccl_single_rank_group(gpu_pkg, nout, monkeypatch):
    """isk_dp_run with a real RCCL communicator (one rank; ISK_DP_FORCE_COLL makes it issue every collective:
    all-reduce(max) / all-gather for the target, all-reduce(sum) of the gradient slabs per step) against isk_run,
    and against the host-driven hook path (ISK_DP_HOST=1)"""
    import importlib
    import torch
    import torch.distributed as dist
    pkg = gpu_pkg
    dp = importlib.import_module("isokann_jl_amd.dp")
    monkeypatch.setenv("ISK_DP_FORCE_COLL", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29571")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        N, mb = 160, 32
        data, odata, om = _coords_case(pkg, 16, 40, N, 3, nout, seed=50 + nout)
        rng = np.random.default_rng(6)
        perms = [rng.permutation(N) for _ in range(4)]
        isos = [pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-2, 1e-3), minibatch=mb) for _ in range(3)]
        pkg.run(isos[0], 2, 2, perms=np.stack(perms))
        comm = dp.Comm(device=torch.device("cuda", 0))
        assert comm.on and comm.world == 1
        local = dp.HipLocal(isos[1])
        assert local.comm_init_native(comm)
        native = dp.dp_run(local, comm, 2, 2, mb, rng, perms=perms)
        monkeypatch.setenv("ISK_DP_HOST", "1")
        hosted = dp.dp_run(dp.HipLocal(isos[2]), comm, 2, 2, mb, rng, perms=perms)
        np.testing.assert_allclose(native, isos[0].losses, rtol=2e-5)
        np.testing.assert_allclose(hosted, isos[0].losses, rtol=2e-5)
        for b in isos[1:]:
            np.testing.assert_allclose(b.model.get_params(), isos[0].model.get_params(), atol=2e-6, rtol=2e-4)
        for b in isos:
            b.close()
    finally:
        dist.destroy_process_group()


def test_evaluation_callers(gpu_pkg):
    """koopman on foreign data, validation loss (+ logger), Koopman variance"""
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 18, 50, 120, 4, 1, seed=41)
    vxs, vys = O.synthetic_coords(18, 40, 4, seed=77)
    feat = O.FeaturesPairs(O.random_pairs(18, 50, seed=42))
    oval = O.SimulationData(np.asarray(vxs, np.float64), np.asarray(vys, np.float64), feat)
    val = pkg.SimulationData(vxs, vys, data.featurizer)
    logger = pkg.ValidationLossLogger(val, logevery=2)
    iso = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(), minibatch=0, loggers=[logger])
    oiso = O.Iso(odata, om, O.setup(O.AdamRegularized(), om), minibatch=0)
    np.testing.assert_allclose(pkg.koopman_coords(iso, vys), O.expectation(om, oval.features[1]), atol=2e-5)
    assert pkg.validationloss(iso, val) == pytest.approx(O.validationloss(oiso, oval), rel=2e-4, abs=1e-9)
    assert pkg.koopman_variance(iso) == pytest.approx(O.koopman_variance(oiso), rel=1e-3)
    # rates(iso) (src/iso.jl:339-351): 1-D chi extended by 1 - chi, matrix logarithm of K chi / chi
    q_ref = O.rates(O.chis(oiso), O.koopman(oiso))
    np.testing.assert_allclose(pkg.rates(iso), q_ref, atol=2e-3 * max(1.0, np.abs(q_ref).max()))
    np.testing.assert_allclose(pkg.rates(iso, lagtime=2.0), pkg.rates(iso) / 2.0, rtol=1e-12)
    perms = np.stack([np.arange(120)] * 4)
    pkg.run(iso, 4, 1, perms=perms)
    O.run(oiso, 4, 1, perms=list(perms))
    assert logger.iters == [2, 4] and len(logger.losses) == 2
    assert logger.losses[-1] == pytest.approx(O.validationloss(oiso, oval), rel=1e-3, abs=1e-9)
    x, y = pkg.chi_kchi(iso)
    np.testing.assert_allclose(O.rates(x.astype(np.float64), y.astype(np.float64)),
                               O.rates(O.chis(oiso), O.koopman(oiso)), atol=2e-3)
    iso.close()


@pytest.mark.parametrize("F,N,K,widths,minibatch", [
    (1, 2, 1, [1, 1, 1], 0),            # smallest non-degenerate problem
    (1, 17, 1, [1, 3, 1], 5),           # K = 1, ragged N, ragged minibatch tail
    (3, 33, 7, [3, 2, 1, 1], 100),      # minibatch > N -> full batch (src/iso.jl:180)
    (6, 128, 2, [6, 12, 5], 16),        # ISA, 5-dimensional chi (5! row permutations)
    (2, 100, 5, [2, 2, 1, 1], 100),     # C1 shape
])
def test_edge_shapes_run_matches_oracle(gpu_pkg, F, N, K, widths, minibatch):
    pkg = gpu_pkg
    rng = np.random.default_rng(F * 100 + N)
    xs = rng.uniform(size=(F, N)) + 0.5
    ys = xs[:, None, :] + 0.3 * rng.uniform(size=(F, K, N))
    om = _rand_model(widths, seed=N, ln=F > 1)      # LayerNorm over a single feature is identically zero
    nout = widths[-1]
    tr, otr = (pkg.TransformShiftscale(), O.TransformShiftscale()) if nout == 1 else (pkg.TransformISA(), O.TransformISA())
    iso = pkg.Iso(pkg.SimulationData(xs, ys), model=_mk_model(pkg, om), opt=pkg.NesterovRegularized(1e-2, 1e-3), target=tr,
                  minibatch=minibatch)
    x32, y32 = np.asarray(xs, np.float32).astype(np.float64), np.asarray(ys, np.float32).astype(np.float64)
    oiso = O.Iso(O.SimulationData(x32, y32), om, O.setup(O.NesterovRegularized(1e-2, 1e-3), om), target=otr, minibatch=minibatch)
    perms = np.stack([rng.permutation(N) for _ in range(3)])
    t = pkg.isotarget(iso)
    want = O.isotarget(otr, om, x32, y32)
    scale = max(1.0, np.abs(want).max())
    np.testing.assert_allclose(t, want, atol=5e-4 * scale)
    pkg.run(iso, 3, 1, perms=perms)
    O.run(oiso, 3, 1, perms=list(perms))
    np.testing.assert_allclose(iso.losses, oiso.losses, rtol=5e-3 if nout > 1 else 2e-4)
    np.testing.assert_allclose(pkg.chis(iso), O.chis(oiso), atol=2e-4 if nout > 1 else 2e-5)
    iso.close()


def test_data_slicing_and_reupload(gpu_pkg):
    """iso.data[6:10]-style slicing (src/simulation.jl:135) builds a new resident data set"""
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 12, 30, 50, 3, 1, seed=13)
    sub = data[10:30]
    assert len(sub) == 20
    a = pkg.Iso(sub, model=_mk_model(pkg, om), minibatch=0)
    f = O.FeaturesPairs(O.random_pairs(12, 30, seed=14))
    want = om(f(np.asarray(sub.coords[0], np.float64)))
    np.testing.assert_allclose(pkg.chis(a), want, atol=2e-5)
    a.close()


@pytest.mark.own_fwd_mode
def test_minibatch_is_mutable_and_save_load_keeps_the_engine_configuration(gpu_pkg, tmp_path):
    """Iso.minibatch is a mutable field in the reference (src/iso.jl:10): assigning it changes what run!() does (the engine
    re-plans its epochs); save / load carry this host's engine configuration (seed, forward mode, per-iteration
    featurization, Koopman sample weights), so a reloaded Iso continues with the same arithmetic and shuffle stream"""
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 12, 30, 96, 3, 1, seed=8)
    perms = np.stack([np.random.default_rng(1).permutation(96) for _ in range(2)])
    a = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-2, 1e-3), minibatch=96)
    b = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-2, 1e-3), minibatch=24)
    a.minibatch = 24                                   # run() must now take 4 optimizer steps per epoch, like b
    pkg.run(a, 2, 1, perms=perms)
    pkg.run(b, 2, 1, perms=perms)
    np.testing.assert_array_equal(a.model.get_params(), b.model.get_params())
    assert a.losses == b.losses
    a.close(); b.close()
    rng = np.random.default_rng(3)
    w = rng.uniform(0.5, 1.5, size=(1, 3, 96))
    iso = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-2, 1e-3), minibatch=32, seed=11, forward_mode=3,
                  featurize_every_iteration=True)
    pkg.set_koopman_weights(iso, w)
    pkg.run(iso, 2)
    path = str(tmp_path / "cfg.npz")
    pkg.save(path, iso)
    isol = pkg.load(path)
    assert (isol.seed, isol.forward_mode, isol.featurize_every_iteration) == (11, 3, True)
    np.testing.assert_array_equal(pkg.koopman(isol), pkg.koopman(iso))          # same weights, same forward arithmetic
    iso.close(); isol.close()


def test_save_load_roundtrip_and_data_growth(gpu_pkg, tmp_path):
    """the reference's own ops test: save / load with coords equality, then keep training
    (test/runtests.jl:69-78); plus mergedata + set_data (src/simulation.jl:162-171)"""
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 14, 36, 96, 3, 3, seed=51)
    iso = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-2, 1e-3), minibatch=32, seed=4)
    pkg.run(iso, 3)
    path = str(tmp_path / "iso.npz")
    pkg.save(path, iso)
    isol = pkg.load(path, seed=4)
    assert np.array_equal(iso.data.coords[0], isol.data.coords[0]) and np.array_equal(iso.data.coords[1], isol.data.coords[1])
    assert isol.losses == iso.losses and isinstance(isol.target, pkg.TransformISA)
    np.testing.assert_array_equal(isol.model.get_params(), iso.model.get_params())
    for a, b in zip(isol.get_opt_state(), iso.get_opt_state()):
        np.testing.assert_array_equal(a, b)
    # resumed training continues exactly like the original (same injected shuffles)
    perms = np.stack([np.random.default_rng(9).permutation(96) for _ in range(2)])
    pkg.run(iso, 2, 1, perms=perms)
    pkg.run(isol, 2, 1, perms=perms)
    np.testing.assert_allclose(isol.losses, iso.losses, rtol=1e-6)
    np.testing.assert_allclose(pkg.chis(isol), pkg.chis(iso), atol=1e-6)
    # data growth: append 32 new starts, model and optimizer state stay
    xs2, ys2 = O.synthetic_coords(14, 32, 3, seed=99)
    merged = pkg.mergedata(iso.data, pkg.SimulationData(xs2, ys2, iso.data.featurizer))
    p_before = iso.model.get_params()
    pkg.set_data(iso, merged)
    assert len(iso.data) == 128 and pkg.chis(iso).shape == (3, 128)
    np.testing.assert_array_equal(iso.model.get_params(), p_before)
    pkg.run(iso, 1)
    assert np.isfinite(iso.losses[-1])
    iso.close()
    isol.close()


def _torch_chi_of_coords(om, pairs, x):
    """independent double-precision chi(coords) in torch (pdist -> LayerNorm -> MLP) for autograd"""
    import torch
    if pairs is None:
        a = x
    else:
        c = x.reshape(-1, 3)
        a = torch.linalg.norm(c[[p[0] for p in pairs]] - c[[p[1] for p in pairs]], dim=1)
    if om.layernorm:
        mu = a.mean()
        var = ((a - mu) ** 2).mean()
        if om.ln_mode == "var_eps2":
            a = (a - mu) / torch.sqrt(var + om.ln_eps ** 2)
        elif om.ln_mode == "std_eps":
            a = (a - mu) / (torch.sqrt(var) + om.ln_eps)
        else:
            a = (a - mu) / torch.sqrt(var + om.ln_eps)
        a = a * torch.tensor(om.ln_scale) + torch.tensor(om.ln_bias)
    fn = {"sigmoid": torch.sigmoid, "identity": lambda z: z, "tanh": torch.tanh, "relu": torch.relu,
          "swish": lambda z: z * torch.sigmoid(z)}
    for W, b, act in zip(om.W, om.b, om.acts()):
        a = fn[act](torch.tensor(W) @ a + torch.tensor(b))
    return a


@pytest.mark.parametrize("nout,mode,act,pdist,hidden", [(1, "var_eps2", "sigmoid", True, (11, 5)), (3, "std_eps", "tanh", True, (11, 5)),
                                                          (2, "var_eps2", "swish", False, (11, 5)),
                                                          (2, "var_eps2", "sigmoid", True, (176, 20))])     # wide: layered path
def test_chicoords_pullback_matches_torch_autograd(gpu_pkg, nout, mode, act, pdist, hidden):
    """d chi / d coords through MLP, LayerNorm and pairwise distances"""
    import torch
    pkg = gpu_pkg
    A, F, N = 15, 40, 24
    rng = np.random.default_rng(60 + nout)
    if pdist:
        xs, ys = O.synthetic_coords(A, N, 2, seed=61)
        pairs = O.random_pairs(A, F, seed=62)
        feat = pkg.FeaturesPairs(pairs)
        widths = [F, *hidden, nout]
    else:
        xs = rng.normal(size=(F, N)) + 1
        ys = rng.normal(size=(F, 2, N)) + 1
        pairs, feat = None, pkg.identity
        widths = [F, *hidden, nout]
    om = _rand_model(widths, seed=63, act=act)
    om.ln_mode = mode
    tr = pkg.TransformShiftscale() if nout == 1 else pkg.TransformISA()
    iso = pkg.Iso(pkg.SimulationData(xs, ys, feat), model=_mk_model(pkg, om), target=tr, minibatch=0)
    cot = rng.normal(size=(nout, N))
    got = pkg.chicoords_pullback(iso, xs, cot)
    x32 = np.asarray(xs, np.float32).astype(np.float64)
    want = np.zeros_like(x32)
    for n in range(N):
        xt = torch.tensor(x32[:, n], requires_grad=True)
        (_torch_chi_of_coords(om, pairs, xt) * torch.tensor(cot[:, n])).sum().backward()
        want[:, n] = xt.grad.numpy()
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, atol=2e-5 * scale, rtol=2e-4)
    # Jacobian helper: rows are the gradients of the individual chi components
    J = pkg.dchidx(iso, xs[:, 0])
    assert J.shape == (nout, x32.shape[0])
    for o in range(nout):
        xt = torch.tensor(x32[:, 0], requires_grad=True)
        _torch_chi_of_coords(om, pairs, xt)[o].backward()
        np.testing.assert_allclose(J[o], xt.grad.numpy(), atol=2e-5 * scale, rtol=2e-4)
    iso.close()


def test_weighted_samples_expectation(gpu_pkg):
    """WeightedSamples (Girsanov): expectation = sum_k w f(y_k) / K  (src/data.jl:215)"""
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 12, 30, 70, 5, 1, seed=71)
    iso = pkg.Iso(data, model=_mk_model(pkg, om), minibatch=0)
    w = np.random.default_rng(3).uniform(0.2, 1.8, size=(1, 5, 70))
    pkg.set_koopman_weights(iso, w)
    fy = om(odata.features[1])                                   # (1, 5, 70)
    want = (fy * w).sum(axis=1) / 5
    np.testing.assert_allclose(pkg.koopman(iso), want, atol=2e-5)
    t = pkg.isotarget(iso)
    np.testing.assert_allclose(t, O.shiftscale(want), atol=5e-5)
    pkg.set_koopman_weights(iso, None)
    np.testing.assert_allclose(pkg.koopman(iso), fy.mean(axis=1), atol=2e-5)
    iso.close()


@pytest.mark.parametrize("widths,ln,act,fmode", [
    ([40, 176, 24, 1], True, "sigmoid", 0),        # first hidden width > 160: layered path
    ([40, 176, 24, 1], True, "sigmoid", 1),
    ([64, 96, 300, 300, 2], False, "tanh", 0),     # tail too wide for LDS (C5-like densenet without LayerNorm)
    ([64, 96, 300, 300, 2], False, "swish", 1),
])
def test_wide_networks_layered_path(gpu_pkg, widths, ln, act, fmode):
    """networks the fused kernels cannot hold run layer by layer: forward, gradients, training, run"""
    pkg = gpu_pkg
    om = _rand_model(widths, seed=sum(widths), ln=ln, act=act)
    rng = np.random.default_rng(5)
    N, K = 80, 3
    xs = rng.normal(size=(widths[0], N)) + 1.0
    ys = xs[:, None, :] + 0.3 * rng.normal(size=(widths[0], K, N))
    nout = widths[-1]
    tr, otr = (pkg.TransformShiftscale(), O.TransformShiftscale()) if nout == 1 else (pkg.TransformISA(), O.TransformISA())
    iso = pkg.Iso(pkg.SimulationData(xs, ys), model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-3, 1e-4), target=tr,
                  minibatch=32, forward_mode=fmode)
    x32, y32 = np.asarray(xs, np.float32).astype(np.float64), np.asarray(ys, np.float32).astype(np.float64)
    np.testing.assert_allclose(iso.model(xs), om(x32), atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(pkg.koopman(iso), O.expectation(om, y32), atol=3e-5, rtol=1e-5)
    tgt = np.asarray(rng.uniform(size=(nout, N)), np.float32)
    idx = rng.permutation(N)[:37]
    g, l = pkg.batch_grads(iso, tgt, idx)
    t64 = tgt.astype(np.float64)
    w = O.loss_weights(t64) if nout > 1 else np.ones(1)
    lo, go, _ = O.batch_loss_and_grads(om, x32[:, idx], t64[:, idx], w)
    gflat = np.concatenate([a.reshape(-1, order="F") for a in go])
    assert abs(l - lo) <= 2e-5 * abs(lo)
    np.testing.assert_allclose(g, gflat, atol=3e-5 * np.abs(gflat).max(), rtol=1e-4)
    oiso = O.Iso(O.SimulationData(x32, y32), om, O.setup(O.AdamRegularized(1e-3, 1e-4), om), target=otr, minibatch=32)
    perms = np.stack([rng.permutation(N) for _ in range(3)])
    pkg.run(iso, 3, 1, perms=perms)
    O.run(oiso, 3, 1, perms=list(perms))
    np.testing.assert_allclose(iso.losses, oiso.losses, rtol=2e-3 if nout > 1 else 2e-4)
    np.testing.assert_allclose(pkg.chis(iso), O.chis(oiso), atol=1e-4 if nout > 1 else 3e-5)
    iso.close()


@pytest.mark.own_fwd_mode
@pytest.mark.parametrize("A,F,widths,ln,act,N,K", [
    (20, 100, [100, 200, 64, 1], False, "sigmoid", 70, 3),        # one ring group of features (<= 128): every wait drains
    (24, 130, [130, 176, 2], True, "tanh", 90, 5),                # two groups, the second 2 features long; ONE hidden layer; two outputs
    (40, 300, [300, 200, 136, 1], True, "sigmoid", 1200, 28),     # 263 row blocks on 256 workgroups: the persistent loop's second
                                                                  #   block (features and weights requested a layer early), a
                                                                  #   half-filled last block, a short last feature group
    (60, 700, [700, 512, 300, 500, 1], False, "relu", 150, 4),    # full-width layers, three hidden layers, steady-state ring
])
def test_wide_fused_bf16_forward_shapes(gpu_pkg, monkeypatch, A, F, widths, ln, act, N, K):
    """the fused bf16 forward of wide networks (kernels_fwd_wide.hip; forward_mode 2 on a layered network with pair features):
    chi and K chi against the fp64 oracle at bf16 accuracy, over the shapes its ring / weight-stream bookkeeping distinguishes"""
    monkeypatch.delenv("ISK_FWD_MODE", raising=False)
    pkg = gpu_pkg
    xs, ys = O.synthetic_coords(A, N, K, seed=F)
    pairs = O.random_pairs(A, F, seed=F + 1)
    om = _rand_model(widths, seed=F + 2, ln=ln, act=act)
    feat = O.FeaturesPairs(pairs)
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_mk_model(pkg, om), minibatch=0, forward_mode=2,
                  featurize_every_iteration=True)
    assert iso.get_option("layered") == 1 and iso.get_option("wide_fwd") == 1
    want_chi = om(feat(np.asarray(xs, np.float64)))
    want_ks = O.expectation(om, feat(np.asarray(ys, np.float64)))
    scale = max(1.0, np.abs(want_chi).max())
    chi, ks = pkg.chis(iso), pkg.koopman(iso)
    assert np.abs(chi - want_chi).max() <= 1e-2 * scale
    assert np.abs(ks - want_ks).max() <= 1e-2 * scale
    assert np.abs(chi - want_chi).max() > 1e-7 * scale            # really the bf16 path
    np.testing.assert_array_equal(pkg.koopman(iso), ks)           # and deterministic
    # the evaluation callers on foreign rows take the same kernel: features handed over by the host (isk_forward: the engine's
    # standardisation writes bf16 rows for them too) and coordinates (isk_chicoords)
    n = min(N, 50)
    fx32 = np.asarray(feat(np.asarray(xs[:, :n], np.float64)), np.float32)
    assert np.abs(iso.model(fx32) - want_chi[:, :n]).max() <= 1e-2 * scale
    sub = pkg.chicoords(iso, ys[:, :, :n])
    assert np.abs(sub.mean(axis=1) - want_ks[:, :n]).max() <= 1e-2 * scale
    iso.close()


@pytest.mark.own_fwd_mode
def test_wide_network_switching_the_training_arithmetic_mid_run(gpu_pkg, monkeypatch):
    """An engine on the fused bf16 forward + bf16 training GEMMs does not fold the f32 weight images nobody reads; switching the
    option back to the in-house f32 training kernels (which do read them) must see the CURRENT weights: gradients of the two
    arithmetics on the same parameters agree at bf16 accuracy after the parameters have moved a long way"""
    monkeypatch.delenv("ISK_FWD_MODE", raising=False)
    pkg = gpu_pkg
    A, F, N, K = 20, 100, 256, 3
    xs, ys = O.synthetic_coords(A, N, K, seed=5)
    pairs = O.random_pairs(A, F, seed=6)
    om = _rand_model([F, 200, 64, 1], seed=7, ln=False)
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_mk_model(pkg, om), opt=pkg.AdamRegularized(5e-2, 1e-4),
                  minibatch=64, forward_mode=2, featurize_every_iteration=True, options={"train_bf16": 1})
    assert iso.get_option("wide_fwd") == 1
    p0 = iso.model.get_params()
    rng = np.random.default_rng(3)
    tgt = np.asarray(rng.uniform(size=(1, N)), np.float32)
    for _ in range(3):
        pkg.train_batch(iso, tgt, perm=rng.permutation(N))
    assert np.abs(iso.model.get_params() - p0).max() > 0.1            # 12 Adam steps at lr 5e-2
    # d chi / d coords runs layer by layer on the f32 images as well
    om_now = _rand_model([F, 200, 64, 1], seed=7, ln=False)
    om_now.set_flat(iso.model.get_params().astype(np.float64))
    cot = rng.normal(size=(1, 8))
    got = pkg.chicoords_pullback(iso, xs[:, :8], cot)
    import torch
    x32 = np.asarray(xs[:, :8], np.float32).astype(np.float64)
    want = np.zeros_like(x32)
    for n in range(8):
        xt = torch.tensor(x32[:, n], requires_grad=True)
        (_torch_chi_of_coords(om_now, pairs, xt) * torch.tensor(cot[:, n])).sum().backward()
        want[:, n] = xt.grad.numpy()
    np.testing.assert_allclose(got, want, atol=5e-5 * np.abs(want).max(), rtol=5e-4)
    idx = rng.permutation(N)[:96]
    g16, l16 = pkg.batch_grads(iso, tgt, idx)
    iso.set_option("train_bf16", 0)
    g32, l32 = pkg.batch_grads(iso, tgt, idx)
    assert abs(l16 - l32) <= 2e-2 * abs(l32)
    assert np.abs(g16 - g32).max() <= 3e-2 * np.abs(g32).max()
    om2 = _rand_model([F, 200, 64, 1], seed=7, ln=False)
    om2.set_flat(iso.model.get_params().astype(np.float64))
    fx = O.FeaturesPairs(pairs)(np.asarray(xs, np.float64))
    lo, go, _ = O.batch_loss_and_grads(om2, fx[:, idx], tgt[:, idx].astype(np.float64), np.ones(1))
    gflat = np.concatenate([a.reshape(-1, order="F") for a in go])
    np.testing.assert_allclose(g32, gflat, atol=5e-5 * np.abs(gflat).max(), rtol=2e-4)
    iso.close()


def test_pseudoinv_needs_multidimensional_chi(gpu_pkg):
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 10, 20, 64, 3, 1, seed=5)
    with pytest.raises(ValueError):
        pkg.Iso(data, model=_mk_model(pkg, om), target=pkg.TransformPseudoInv(), minibatch=0)


# --------------------------------------------------------------- plain bf16 forward (forward_mode 2, C5's arithmetic) ----
@pytest.mark.own_fwd_mode
@pytest.mark.parametrize("widths,ln", [([231, 38, 6, 1], True), ([100, 150, 20, 3], True), ([64, 96, 300, 300, 2], False),
                                       ([40, 512, 40, 1], False)])
def test_forward_plain_bf16_within_1e2(gpu_pkg, widths, ln, monkeypatch):
    """forward_mode 2: bf16 operands (leading plane only), f32 accumulate -- SURVEY 8d: 'bf16 path: stated separately
    (expected ~1e-2)'.  Covers the fused kernel, the layered dense-block path and a 512-wide layer (two 16-tile launches)."""
    monkeypatch.delenv("ISK_FWD_MODE", raising=False)
    pkg = gpu_pkg
    om = _rand_model(widths, seed=sum(widths) + 1, ln=ln)
    rng = np.random.default_rng(8)
    N, K = 96, 3
    xs = rng.normal(size=(widths[0], N)) + 1
    ys = rng.normal(size=(widths[0], K, N)) + 1
    tr = pkg.TransformShiftscale() if widths[-1] == 1 else pkg.TransformISA()
    out = {}
    for mode in (0, 2):
        iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.identity), model=_mk_model(pkg, om), target=tr, minibatch=0, forward_mode=mode)
        out[mode] = (pkg.chis(iso), pkg.koopman(iso))
        iso.close()
    want = om(np.asarray(xs, np.float32).astype(np.float64))
    scale = max(1.0, np.abs(want).max())
    assert np.abs(out[0][0] - want).max() <= 2e-5 * scale                 # the f32 path, for reference
    assert np.abs(out[2][0] - want).max() <= 1e-2 * scale                 # bf16 operands
    assert np.abs(out[2][0] - want).max() > 1e-6 * scale                  # ... and it really is the reduced-precision path
    assert np.abs(out[2][1] - out[0][1]).max() <= 1e-2 * scale


@pytest.mark.own_fwd_mode
def test_run_plain_bf16_forward_tracks_f32(gpu_pkg, monkeypatch):
    """three iterations with the plain-bf16 Koopman forward (training stays f32) stay close to the f32 run"""
    monkeypatch.delenv("ISK_FWD_MODE", raising=False)
    pkg = gpu_pkg
    data, odata, om = _coords_case(pkg, 18, 60, 200, 4, 1, seed=77)
    perms = np.stack([np.arange(200)] * 3)
    res = {}
    for mode in (0, 2):
        iso = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(), minibatch=0, forward_mode=mode)
        pkg.run(iso, 3, 1, perms=perms)
        res[mode] = (np.asarray(iso.losses), pkg.chis(iso))
        iso.close()
    np.testing.assert_allclose(res[2][0], res[0][0], rtol=0.1)
    assert np.abs(res[2][1] - res[0][1]).max() <= 2e-2


# --------------------------------------------------------------- every template instantiation of the bf16 forward ----
@pytest.mark.own_fwd_mode
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_bf16_forward_every_tile_count(gpu_pkg, mode, monkeypatch):
    """first hidden widths 5, 21, ..., 149 (1..10 output tiles: every instantiation of the fused split-bf16 / plain-bf16 /
    split-f16 kernel) and wide layers whose dense blocks hit every tile count of the layered path (1..8 split, 1..16
    plain; split-f16 requests run the split-bf16 kernels there)"""
    monkeypatch.delenv("ISK_FWD_MODE", raising=False)
    pkg = gpu_pkg
    rng = np.random.default_rng(12)
    F, N, K = 70, 40, 2
    xs = rng.normal(size=(F, N)) + 1
    ys = rng.normal(size=(F, K, N)) + 1
    tol = 1e-2 if mode == 2 else 2e-5
    fused = [[F, 16 * k + 5, 7, 1] for k in range(10)]
    dense = [[F, 160 + 16 * k + 3, 9, 1] for k in range(1, 17)]         # 171..419 wide: blocks of (8|16) tiles + remainder
    for widths in fused + dense:
        om = _rand_model(widths, seed=widths[1], ln=True)
        iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.identity), model=_mk_model(pkg, om), minibatch=0, forward_mode=mode)
        want = om(np.asarray(xs, np.float32).astype(np.float64))
        got = pkg.chis(iso)
        assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), widths
        kw = O.expectation(om, np.asarray(ys, np.float32).astype(np.float64))
        assert np.abs(pkg.koopman(iso) - kw).max() <= tol * max(1.0, np.abs(kw).max()), widths
        iso.close()

@pytest.mark.gpu
@pytest.mark.parametrize("nout", [1, 3])
def test_dp_native_engine_drawn_shuffles_match_isk_run(gpu_pkg, nout):
    """isk_dp_run with perms = NULL draws the shuffle orders from the engine's seeded generator, one per epoch, exactly
    like isk_run does: with one rank and no communicator the two runs are the same computation"""
    import importlib
    pkg = gpu_pkg
    dp = importlib.import_module("isokann_jl_amd.dp")
    N, mb = 160, 32
    data, odata, om = _coords_case(pkg, 16, 40, N, 3, nout, seed=70 + nout)
    isos = [pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-2, 1e-3), minibatch=mb) for _ in range(2)]
    pkg.run(isos[0], 3, 2)                                   # engine-drawn shuffles
    comm = dp.Comm(device=None)
    assert not comm.on and comm.world == 1
    local = dp.HipLocal(isos[1])
    losses = dp.dp_run(local, comm, 3, 2, mb, None)          # rng None -> the engine draws
    assert len(losses) == 6 and np.all(np.isfinite(losses))
    np.testing.assert_allclose(losses, isos[0].losses, rtol=2e-5)
    np.testing.assert_allclose(isos[1].model.get_params(), isos[0].model.get_params(), atol=2e-6, rtol=2e-4)
    for b in isos:
        b.close()

