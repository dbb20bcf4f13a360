"""Round-4 additions at the boundary: data growth without re-upload (isk_append_data), the coordinate pullback at villin shape,
and the documented shape limits as error codes."""
import numpy as np
import pytest

from oracle import isokann_oracle as O
from tests import benchshape_cases as bc

pytestmark = [pytest.mark.gpu, pytest.mark.own_fwd_mode, pytest.mark.usefixtures("fwd_mode")]


def _model(pkg, widths, seed, ln=True):
    rng = np.random.default_rng(seed)
    om = O.glorot_model(widths, rng, layernorm=ln)
    for b in om.b:
        b[...] = 0.1 * rng.normal(size=b.shape)
    m = pkg.densenet(om.widths, layernorm=ln)
    m.params = om.get_flat().astype(np.float32)
    return om, m


@pytest.mark.parametrize("every_iter", [False, True])
def test_append_data_equals_a_full_upload(gpu_pkg, every_iter):
    """addcoords! / the cutoff window of run_kde! (src/iso.jl:238,288-290; src/simulation.jl:162-185): `iso.data = ...` with data
    that grew at the end and lost start points at the front keeps the resident part on the device (isk_append_data).  The engine
    must then be in exactly the state a full isk_set_data of the same data leaves: chi, K chi and a training epoch bit for bit."""
    pkg = gpu_pkg
    A, F, K, nout = 18, 70, 4, 2
    xs, ys = O.synthetic_coords(A, 300, K, seed=2)
    pairs = O.random_pairs(A, F, seed=3)
    om, _ = _model(pkg, [F, 20, 6, nout], seed=4)
    mk = lambda x, y: pkg.SimulationData(x, y, pkg.FeaturesPairs(pairs))
    kw = dict(opt=pkg.AdamRegularized(), minibatch=64, target=pkg.TransformISA(), featurize_every_iteration=every_iter)
    grown = pkg.Iso(mk(xs[:, :200], ys[:, :, :200]), model=_model(pkg, [F, 20, 6, nout], seed=4)[1], **kw)
    # (1) addcoords!: 60 new start points behind the resident 200
    pkg.set_data(grown, mk(xs[:, :260], ys[:, :, :260]))
    assert grown.data.coords[0].shape[1] == 260
    # (2) the cutoff window: the last 180 of them ...   (3) ... and both at once through the explicit entry
    pkg.set_data(grown, mk(xs[:, 80:260], ys[:, :, 80:260]))
    pkg.append_data(grown, xs[:, 260:300], ys[:, :, 260:300], drop_first=20)
    np.testing.assert_array_equal(grown.data.coords[0], np.asarray(xs[:, 100:300], np.float32))
    full = pkg.Iso(mk(xs[:, 100:300], ys[:, :, 100:300]), model=_model(pkg, [F, 20, 6, nout], seed=4)[1], **kw)
    np.testing.assert_array_equal(pkg.chis(grown), pkg.chis(full))
    np.testing.assert_array_equal(pkg.koopman(grown), pkg.koopman(full))
    perms = np.stack([np.random.default_rng(7).permutation(200) for _ in range(2)]).astype(np.int32)
    pkg.run(grown, 2, 1, perms=perms)
    pkg.run(full, 2, 1, perms=perms)
    assert grown.losses == full.losses
    np.testing.assert_array_equal(grown.model.get_params(), full.model.get_params())
    # error codes: nothing left, mismatching K
    with pytest.raises(pkg.capi.IskError) as ei:
        pkg.append_data(grown, None, None, drop_first=200)
    assert ei.value.status == pkg.capi.EINVAL
    with pytest.raises(ValueError):
        pkg.append_data(grown, xs[:, :5], ys[:, :3, :5])
    grown.close(); full.close()


def test_villin_shape_pullback_matches_torch_autograd(gpu_pkg):
    """d chi / d coords (the counterpart of the reference's sqpairdist_bwd_kernel!, src/utils/pairdists.jl:153-196, through
    LayerNorm and the MLP) at the VILLIN shape -- 591 atoms, the bench's 1700 pairs, [1700,142,12,3]+LN -- for eight columns
    against torch's fp64 autograd of the same network on the float32 coordinates: 2e-5 of the largest gradient entry"""
    import torch
    pkg = gpu_pkg
    A, F, nout, ncol = bc.C3_ATOMS, bc.C3_F, 3, 8
    xs, ys = O.synthetic_coords(A, 32, 2, seed=5)
    pairs = O.random_pairs(A, F, seed=0)
    om, m = _model(pkg, [F, 142, 12, nout], seed=21)
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=m, target=pkg.TransformISA(), minibatch=0)
    cot = np.random.default_rng(1).normal(size=(nout, ncol))
    got = pkg.chicoords_pullback(iso, xs[:, :ncol], cot)
    x32 = np.asarray(xs[:, :ncol], np.float32).astype(np.float64)
    P = torch.tensor(np.asarray(pairs))
    want = np.zeros_like(x32)
    for n in range(ncol):
        xt = torch.tensor(x32[:, n], requires_grad=True)
        c = xt.reshape(-1, 3)
        a = torch.sqrt(((c[P[:, 0]] - c[P[:, 1]]) ** 2).sum(dim=1))
        mu = a.mean()
        var = ((a - mu) ** 2).mean()
        a = (a - mu) / torch.sqrt(var + om.ln_eps ** 2) * torch.tensor(om.ln_scale) + torch.tensor(om.ln_bias)
        for W, b, act in zip(om.W, om.b, om.acts()):
            z = torch.tensor(W) @ a + torch.tensor(b)
            a = torch.sigmoid(z) if act == "sigmoid" else z
        (a * torch.tensor(cot[:, n])).sum().backward()
        want[:, n] = xt.grad.numpy()
    scale = np.abs(want).max()
    assert scale > 0
    np.testing.assert_allclose(got, want, atol=2e-5 * scale, rtol=2e-4)
    iso.close()


def test_documented_shape_limits_are_error_codes(gpu_pkg):
    """what the kernels do not cover is refused with ISK_EUNSUPPORTED at isk_create / at the call, never computed wrongly:
    TransformISA beyond 64 outputs (up to 8 the device chain, up to 64 the host solve: test_isa_with_more_than_eight_outputs), a
    layer wider than 16384, more than 10240 features + coordinates per sample beyond the register featurizer's range (src/utils/pairdists.jl has no such bound)"""
    pkg = gpu_pkg
    rng = np.random.default_rng(0)
    xs, ys = rng.normal(size=(12, 40)), rng.normal(size=(12, 2, 40))
    with pytest.raises(pkg.capi.IskError) as ei:
        pkg.Iso(pkg.SimulationData(xs, ys, pkg.identity), model=pkg.densenet([12, 80, 65], layernorm=True), target=pkg.TransformISA(), minibatch=0)
    assert ei.value.status == pkg.capi.EUNSUPPORTED
    with pytest.raises(pkg.capi.IskError) as ei:
        pkg.Iso(pkg.SimulationData(xs, ys, pkg.identity), model=pkg.densenet([12, 16400, 1], layernorm=False), minibatch=0)
    assert ei.value.status == pkg.capi.EUNSUPPORTED
    # beyond the register featurizer's range (2048 features / 682 atoms) a sample's coordinates and feature row are staged in LDS: more than
    # 10240 of them together are refused at isk_create (round 5: up to there forward, training AND the coordinate pullback run --
    # tests/test_gpu_pullback_wide.py; before, the pullback stopped at 2048 features and larger rows failed at the first launch)
    F = 10300
    xw, yw = rng.normal(size=(F, 4)), rng.normal(size=(F, 2, 4))
    with pytest.raises(pkg.capi.IskError) as ei:
        pkg.Iso(pkg.SimulationData(xw, yw, pkg.identity), model=pkg.densenet([F, 24, 4, 1], layernorm=True), minibatch=0)
    assert ei.value.status == pkg.capi.EUNSUPPORTED
    F = 9000
    xw, yw = rng.normal(size=(F, 4)), rng.normal(size=(F, 2, 4))
    iso = pkg.Iso(pkg.SimulationData(xw, yw, pkg.identity), model=pkg.densenet([F, 24, 4, 1], layernorm=True), minibatch=0)
    assert np.all(np.isfinite(pkg.chis(iso)))
    g = pkg.chicoords_pullback(iso, xw[:, :2], np.ones((1, 2)))
    assert g.shape == (F, 2) and np.all(np.isfinite(g)) and np.abs(g).max() > 0
    iso.close()


@pytest.mark.parametrize("widths,ln,mb,opt", [([60, 13, 4, 1], True, 32, "adam"), ([60, 13, 4, 1], True, 50, "nesterov"), ([231, 38, 6, 1], True, 100, "adam"),
                                              ([200, 142, 12, 3], True, 100, "nesterov"), ([200, 40, 3], False, 64, "adam"),
                                              ([100, 20, 10, 5, 2], True, 64, "adam"), ([48, 150, 40, 2], True, 16, "adam")])
def test_optimizer_step_riding_on_the_gradient_kernel(gpu_pkg, widths, ln, mb, opt):
    """option fuse_opt (one row slab, one GPU): train_grad1n applies WeightDecay + Adam / Nesterov to the parameters whose gradient
    it produces -- three launches per optimizer step instead of four.  Same gradients, same update formula: the parameters and
    the optimizer state must come out BIT-identical to the unfused chain (only the batch loss is summed in another order), over
    narrow and generic tails, with and without LayerNorm, and the run must agree with the oracle as before.  The last case's small-
    gradient vector has more elements than the grid has 16-lane groups: there the launcher must fall back to four launches."""
    pkg = gpu_pkg
    F, nout, N = widths[0], widths[-1], 320
    A = 24
    xs, ys = O.synthetic_coords(A, N, 2, seed=11)
    pairs = O.random_pairs(A, F, seed=12)
    om, _ = _model(pkg, widths, seed=13, ln=ln)
    st = O.setup(O.OptRule(opt, 1e-2, 1e-3), om)
    odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64), O.FeaturesPairs(pairs))
    isos = {f: pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_model(pkg, widths, seed=13, ln=ln)[1],
                       opt=pkg.OptSpec(opt, 1e-2, 1e-3), minibatch=mb, options={"fuse_opt": f}) for f in (1, 0)}
    rng = np.random.default_rng(0)
    for ep in range(3):
        tgt = np.asarray(rng.uniform(size=(nout, N)), np.float32)
        perm = rng.permutation(N)
        lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), st, mb, perm=perm)
        l1, l0 = (pkg.train_batch(isos[f], tgt, perm=perm) for f in (1, 0))
        assert abs(l1 - l0) <= 1e-6 * abs(l0) and abs(l1 - lo) <= 1e-4 * abs(lo), (ep, l1, l0, lo)
    np.testing.assert_array_equal(isos[1].model.get_params(), isos[0].model.get_params())
    m1, v1, bt1 = isos[1].get_opt_state()
    m0, v0, bt0 = isos[0].get_opt_state()
    np.testing.assert_array_equal(m1, m0)
    np.testing.assert_array_equal(v1, v0)
    assert bt1[0] == bt0[0] and bt1[1] == bt0[1]
    np.testing.assert_allclose(isos[1].model.get_params(), om.get_flat(), atol=3e-5, rtol=3e-4)
    # a non-finite loss: nothing is applied, the epoch reports the collapse
    before = isos[1].model.get_params()
    with pytest.raises(pkg.DomainError) as ei:
        pkg.train_batch(isos[1], np.full((nout, N), np.nan, np.float32), perm=np.arange(N))
    assert ei.value.status == pkg.capi.ECOLLAPSED
    np.testing.assert_array_equal(isos[1].model.get_params(), before)
    for iso in isos.values():
        iso.close()


@pytest.mark.parametrize("widths,ln,mb,N", [([60, 13, 4, 1], True, 32, 320), ([60, 13, 4, 1], True, 50, 330), ([231, 38, 6, 1], True, 100, 320),
                                            ([200, 142, 12, 3], True, 0, 1000), ([200, 142, 12, 3], True, 0, 1016), ([200, 40, 3], False, 64, 320),
                                            ([100, 20, 10, 5, 2], True, 64, 200), ([1700, 142, 12, 3], True, 1000, 2000)])
def test_tail_step_riding_on_the_layer1_forward(gpu_pkg, experimental_lib, widths, ln, mb, N):
    """option fuse_tail (narrow tail, pairnet default activations): train_fwd1n and the tail step are ONE launch -- the workgroup
    whose K-slice partial of a 32-row block arrives last runs the tail of those rows (kernels_train_fwdtail.hip).  The same partials
    are summed in the same order by the same device code, so losses, parameters and optimizer state must be BIT-identical to the
    two launches; cases: one / two / three tail layers, no LayerNorm, a batch whose unit count is odd (50 rows = 4 units: no; 1000
    rows = 63 units and 1016 = 64: the full-batch cases), several K slices (F = 1700 at B = 1000: the bench's step), and repeated
    epochs (the arrival counters must be back at zero after every launch)."""
    pkg = gpu_pkg
    F, nout = widths[0], widths[-1]
    A = 24 if F <= 231 else 60
    xs, ys = O.synthetic_coords(A, N, 2, seed=11)
    pairs = O.random_pairs(A, F, seed=12)
    om, _ = _model(pkg, widths, seed=13, ln=ln)
    st = O.setup(O.OptRule("adam", 1e-2, 1e-3), om)
    odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64), O.FeaturesPairs(pairs))
    isos = {f: pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_model(pkg, widths, seed=13, ln=ln)[1],
                       opt=pkg.OptSpec("adam", 1e-2, 1e-3), minibatch=mb, options={"fuse_tail": f}) for f in (1, 0)}
    rng = np.random.default_rng(0)
    for ep in range(4):
        tgt = np.asarray(rng.uniform(size=(nout, N)), np.float32)
        perm = rng.permutation(N)
        lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), st, mb, perm=perm)
        l1, l0 = (pkg.train_batch(isos[f], tgt, perm=perm) for f in (1, 0))
        assert l1 == l0 and abs(l1 - lo) <= 1e-4 * abs(lo), (ep, l1, l0, lo)
        assert isos[1].get_option("fuse_tail_active") == 1 and isos[0].get_option("fuse_tail_active") == 0
    np.testing.assert_array_equal(isos[1].model.get_params(), isos[0].model.get_params())
    m1, v1, bt1 = isos[1].get_opt_state()
    m0, v0, bt0 = isos[0].get_opt_state()
    np.testing.assert_array_equal(m1, m0)
    np.testing.assert_array_equal(v1, v0)
    assert bt1[0] == bt0[0] and bt1[1] == bt0[1]
    np.testing.assert_allclose(isos[1].model.get_params(), om.get_flat(), atol=3e-5, rtol=3e-4)
    for iso in isos.values():
        iso.close()


@pytest.mark.parametrize("train_bf16", [0, 1])
@pytest.mark.parametrize("ln", [False, True])
def test_wide_network_epoch_images_gathered_up_front(gpu_pkg, experimental_lib, train_bf16, ln):
    """wide networks (layered path), option epoch_gather: the batch images of every batch of an epoch are written by one launch in
    front of the epoch's steps instead of one launch per step.  Same images, same products: losses, parameters and optimizer state
    bit-identical to the per-step gather, in Float32 and in bf16 arithmetic, over epochs with different shuffle orders; and a
    gradient-only call between epochs (rows that are not a batch of an epoch) is unaffected."""
    pkg = gpu_pkg
    widths, N, mb = [64, 200, 48, 2], 330, 64
    A = 24
    xs, ys = O.synthetic_coords(A, N, 2, seed=21)
    pairs = O.random_pairs(A, widths[0], seed=22)
    isos = {g: pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_model(pkg, widths, seed=23, ln=ln)[1],
                       opt=pkg.OptSpec("adam", 1e-2, 1e-3), minibatch=mb, options={"epoch_gather": g, "train_bf16": train_bf16}) for g in (1, 0)}
    assert all(i.get_option("layered") == 1 for i in isos.values())
    rng = np.random.default_rng(0)
    for ep in range(3):
        tgt = np.asarray(rng.uniform(size=(widths[-1], N)), np.float32)
        perm = rng.permutation(N)
        l1, l0 = (pkg.train_batch(isos[g], tgt, perm=perm) for g in (1, 0))
        assert l1 == l0 and np.isfinite(l1), (ep, l1, l0)
    np.testing.assert_array_equal(isos[1].model.get_params(), isos[0].model.get_params())
    m1, v1, _ = isos[1].get_opt_state()
    m0, v0, _ = isos[0].get_opt_state()
    np.testing.assert_array_equal(m1, m0)
    np.testing.assert_array_equal(v1, v0)
    for iso in isos.values():
        iso.close()


@pytest.mark.parametrize("train_bf16", [0, 1])
@pytest.mark.parametrize("opt", ["adam", "nesterov"])
def test_wide_network_step_boundary_as_one_launch(gpu_pkg, train_bf16, opt):
    """wide networks without a leading LayerNorm, option fuse_boundary: the optimizer step, the next step's weight images and the next
    step's batch images are ONE launch (opt_prep_kernel) instead of opt_step + prep_images.  The update rule is the shared opt_rule
    and the images are written from the values opt_step would have stored: losses, parameters and optimizer state bit-identical to
    the two launches, over several epochs (the first step of an epoch prepares its own images, the last one gathers nothing); a
    non-finite loss applies nothing and reports the collapse."""
    pkg = gpu_pkg
    widths, N, mb = [64, 200, 48, 2], 330, 64
    A = 24
    xs, ys = O.synthetic_coords(A, N, 2, seed=31)
    pairs = O.random_pairs(A, widths[0], seed=32)
    isos = {f: pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_model(pkg, widths, seed=33, ln=False)[1],
                       opt=pkg.OptSpec(opt, 1e-2, 1e-3), minibatch=mb, options={"fuse_boundary": f, "train_bf16": train_bf16}) for f in (1, 0)}
    assert all(i.get_option("layered") == 1 for i in isos.values())
    rng = np.random.default_rng(0)
    for ep in range(3):
        tgt = np.asarray(rng.uniform(size=(widths[-1], N)), np.float32)
        perm = rng.permutation(N)
        l1, l0 = (pkg.train_batch(isos[f], tgt, perm=perm) for f in (1, 0))
        assert l1 == l0 and np.isfinite(l1), (ep, l1, l0)
        np.testing.assert_array_equal(pkg.chis(isos[1]), pkg.chis(isos[0]))          # (a forward between epochs: the images it needs are rebuilt)
    np.testing.assert_array_equal(isos[1].model.get_params(), isos[0].model.get_params())
    m1, v1, _ = isos[1].get_opt_state()
    m0, v0, _ = isos[0].get_opt_state()
    np.testing.assert_array_equal(m1, m0)
    np.testing.assert_array_equal(v1, v0)
    before = isos[1].model.get_params()
    for f in (1, 0):
        with pytest.raises(pkg.DomainError) as ei:
            pkg.train_batch(isos[f], np.full((widths[-1], N), np.nan, np.float32), perm=np.arange(N))
        assert ei.value.status == pkg.capi.ECOLLAPSED
        np.testing.assert_array_equal(isos[f].model.get_params(), before)
    # ... and the engine trains on afterwards (the images were left valid)
    tgt = np.asarray(rng.uniform(size=(widths[-1], N)), np.float32)
    perm = rng.permutation(N)
    l1, l0 = (pkg.train_batch(isos[f], tgt, perm=perm) for f in (1, 0))
    assert l1 == l0
    for iso in isos.values():
        iso.close()


@pytest.mark.parametrize("widths,N,mb", [([64, 200, 48, 2], 330, 64), ([96, 256, 136, 1], 300, 100), ([64, 180, 64, 40, 3], 200, 0)])
def test_wide_network_backward_head_as_one_launch(gpu_pkg, widths, N, mb):
    """wide networks with >= 3 layers and <= 4 outputs, option fuse_head: the last layer's forward, the loss, delta_L and delta_{L-1}
    are ONE launch (head_kernel) and the next launch sums its per-workgroup partials of dW_L / db_L / the loss.  Float32 on the
    master weights, another summation order than the three launches it replaces: gradients, losses and trained parameters must agree
    with them to float32 rounding (Float32 arithmetic) and with the oracle as before; in bf16 arithmetic the head is the more exact
    of the two (no bf16 rounding of delta_L and of the last layer's weights), so both are held to the oracle at the bf16 tolerance."""
    pkg = gpu_pkg
    A = 24
    xs, ys = O.synthetic_coords(A, N, 2, seed=41)
    pairs = O.random_pairs(A, widths[0], seed=42)
    nout = widths[-1]
    rng = np.random.default_rng(0)
    tgt = np.asarray(rng.uniform(size=(nout, N)), np.float32)
    for train_bf16 in (0, 1):
        om, _ = _model(pkg, widths, seed=43, ln=False)
        st = O.setup(O.OptRule("adam", 1e-2, 1e-3), om)
        odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64), O.FeaturesPairs(pairs))
        isos = {h: pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_model(pkg, widths, seed=43, ln=False)[1],
                           opt=pkg.OptSpec("adam", 1e-2, 1e-3), minibatch=mb, options={"fuse_head": h, "train_bf16": train_bf16}) for h in (1, 0)}
        assert all(i.get_option("layered") == 1 for i in isos.values())
        idx = rng.permutation(N)[:min(N, 100)]
        g1, gl1 = pkg.batch_grads(isos[1], tgt, idx)
        g0, gl0 = pkg.batch_grads(isos[0], tgt, idx)
        scale = np.max(np.abs(g0))
        tol = 2e-2 if train_bf16 else 2e-6
        assert abs(gl1 - gl0) <= (1e-2 if train_bf16 else 1e-6) * abs(gl0), (gl1, gl0)
        np.testing.assert_allclose(g1, g0, atol=tol * scale, rtol=0)
        for ep in range(2):
            perm = rng.permutation(N)
            lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), st, mb, perm=perm)
            l1, l0 = (pkg.train_batch(isos[h], tgt, perm=perm) for h in (1, 0))
            assert abs(l1 - lo) <= (5e-2 if train_bf16 else 1e-4) * abs(lo) and abs(l0 - lo) <= (5e-2 if train_bf16 else 1e-4) * abs(lo), (ep, l1, l0, lo)
        if not train_bf16:
            np.testing.assert_allclose(isos[1].model.get_params(), om.get_flat(), atol=5e-5, rtol=5e-4)
        for iso in isos.values():
            iso.close()


@pytest.mark.parametrize("d,every_iter", [(9, False), (12, True)])
def test_isa_with_more_than_eight_outputs(gpu_pkg, d, every_iter):
    """TransformISA takes any number of outputs in the reference (src/isotarget.jl:100-127; fixperm walks d! permutations,
    "TODO: use the hungarian algorithm for larger systems").  Up to 8 the target runs on the device; beyond, chi and K chi come to
    the host for the Float64 solve with fixperm as an assignment problem.  At a random initialisation a 9- or 12-vertex simplex is
    badly conditioned (the forward's 2e-7 is amplified beyond any tolerance), so the stages are checked on their own: the target
    against the oracle's Float64 solve evaluated on the engine's OWN chi / K chi (as the full-N C3 test does), its known answer,
    and one training epoch on that target against the oracle's train_batch!."""
    pkg = gpu_pkg
    from tests.test_gpu_parity import _coords_case, _mk_model
    N = 320
    data, odata, om = _coords_case(pkg, 24, 120, N, 3, d, seed=70 + d)
    iso = pkg.Iso(data, model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-3, 1e-4), target=pkg.TransformISA(), minibatch=64,
                  featurize_every_iteration=every_iter)
    np.testing.assert_allclose(pkg.chis(iso), om(odata.features[0]), atol=2e-5)
    t = pkg.isotarget(iso)
    k64, c64 = pkg.koopman(iso).astype(np.float64), pkg.chis(iso).astype(np.float64)
    want = O.fixperm(O.myisa(k64.T).T @ k64, c64)
    assert np.abs(t - want).max() <= 2e-5 * np.abs(want).max()
    sel = t[:, O.indexmap(k64.T)]                                   # the simplex columns are unit vectors (in some row order)
    scale = np.abs(want).max()
    assert np.allclose(np.sort(sel, axis=0)[-1], 1.0, atol=1e-5 * scale) and np.allclose(np.abs(sel).sum(axis=0), 1.0, atol=1e-4 * scale)
    # one epoch (five optimizer steps of 64 rows) on that target
    perm = np.random.default_rng(d).permutation(N)
    st = O.setup(O.AdamRegularized(1e-3, 1e-4), om)
    lo = O.train_batch(om, odata.features[0], t.astype(np.float64), st, 64, perm=perm)
    pkg.run(iso, 1, 1, perms=perm[None, :])
    np.testing.assert_allclose(iso.losses[-1], lo, rtol=2e-4)
    np.testing.assert_allclose(pkg.chis(iso), om(odata.features[0]), atol=5e-5)
    iso.close()
