"""Wide hidden layers on the layered path (the reference's densenet takes any widths, src/models.jl:87-92; isk_create accepts up to 16384
since round 5 -- until then 4096, of which 2560 .. 4096 failed at the first fold launch: its transpose tile outgrew the LDS):
forward, one epoch of train_batch! and the parameters behind it against the oracle."""
import numpy as np
import pytest

from oracle import isokann_oracle as O
from tests.test_gpu_parity import _mk_model, _rand_model

pytestmark = [pytest.mark.gpu]


@pytest.mark.parametrize("widths", [[64, 1500, 2], [64, 4096, 2], [40, 2100, 1100, 1], [64, 5000, 2], [40, 4200, 4100, 1]])
def test_wide_hidden_layers(gpu_pkg, widths):
    pkg = gpu_pkg
    rng = np.random.default_rng(0)
    F, nout, N, K = widths[0], widths[-1], 96, 2
    xs, ys = rng.normal(size=(F, N)), rng.normal(size=(F, K, N))
    om = _rand_model(widths, seed=3)
    odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64))
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.identity), model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-3, 1e-4), minibatch=32,
                  target=pkg.TransformISA() if nout > 1 else None)
    want = O.forward(om, odata.features[0])
    np.testing.assert_allclose(pkg.chis(iso), want, atol=5e-5)
    tgt = np.asarray(rng.uniform(size=(nout, N)), np.float32)
    st = O.setup(O.AdamRegularized(1e-3, 1e-4), om)
    perm = rng.permutation(N)
    lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), st, 32, perm=perm)
    l = pkg.train_batch(iso, tgt, perm=perm)
    assert abs(l - lo) <= 2e-4 * abs(lo), (l, lo)
    # (Adam divides by sqrt(v): where the second moment is at rounding level a step's direction is not determined -- a handful of
    # the 3e5 parameters, as in tests/test_gpu_epoch_persist.py)
    got, ref = iso.model.get_params(), om.get_flat()
    bad = np.abs(got - ref) > 3e-5 + 3e-4 * np.abs(ref)
    assert bad.mean() < 1e-4, bad.mean()
    assert np.abs(got - ref).max() < 2e-3
    iso.close()


def test_wide_hidden_layer_bf16_training(gpu_pkg):
    """the same beyond 4096 on the bf16 training GEMMs (option train_bf16) and the plain-bf16 forward: bf16 tolerances"""
    pkg = gpu_pkg
    rng = np.random.default_rng(1)
    widths, N, K = [64, 4500, 300, 1], 128, 2
    xs, ys = rng.normal(size=(64, N)), rng.normal(size=(64, K, N))
    om = _rand_model(widths, seed=4)
    odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64))
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.identity), model=_mk_model(pkg, om), opt=pkg.AdamRegularized(1e-3, 1e-4), minibatch=64,
                  forward_mode=2, options={"train_bf16": 1})
    want = O.forward(om, odata.features[0])
    np.testing.assert_allclose(pkg.chis(iso), want, atol=3e-2 * max(1.0, np.abs(want).max()))
    tgt = np.asarray(rng.uniform(size=(1, N)), np.float32)
    st = O.setup(O.AdamRegularized(1e-3, 1e-4), om)
    perm = rng.permutation(N)
    lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), st, 64, perm=perm)
    l = pkg.train_batch(iso, tgt, perm=perm)
    assert abs(l - lo) <= 3e-2 * abs(lo), (l, lo)
    iso.close()
