"""Degenerate inputs end in the reference's DomainErrors (or in finite results), never in a crash, a hang or silent garbage: NaN / Inf /
1e30 coordinates, N identical samples, atoms on top of each other (zero distances: LayerNorm over a constant row, the pullback's
subgradient at 0), a one-sample batch."""
import numpy as np
import pytest

from oracle import isokann_oracle as O
from tests.test_gpu_parity import _mk_model, _rand_model

pytestmark = [pytest.mark.gpu]


def _iso(pkg, xs, ys, A, F, nout, mb=0):
    pairs = O.random_pairs(A, F, seed=1)
    om = _rand_model([F, 20, 6, nout], seed=2)
    return pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_mk_model(pkg, om), minibatch=mb,
                   target=pkg.TransformISA() if nout > 1 else None, featurize_every_iteration=True)


@pytest.mark.parametrize("nout", [1, 3])
@pytest.mark.parametrize("poison", ["nan", "inf", "huge"])
def test_poisoned_coordinates_raise_a_domain_error(gpu_pkg, nout, poison):
    pkg = gpu_pkg
    A, F, N, K = 12, 40, 64, 3
    xs, ys = O.synthetic_coords(A, N, K, seed=3)
    xs, ys = np.array(xs, np.float32), np.array(ys, np.float32)
    v = {"nan": np.nan, "inf": np.inf, "huge": 1e30}[poison]
    ys[5, 1, 7] = v
    xs[3, 9] = v
    iso = _iso(pkg, xs, ys, A, F, nout)
    try:
        with pytest.raises(pkg.DomainError):
            pkg.run(iso, 2, 1)
        # the engine is still usable: clean data, a finite iteration
        xs2, ys2 = O.synthetic_coords(A, N, K, seed=4)
        pkg.set_data(iso, pkg.SimulationData(xs2, ys2, iso.data.featurizer))
        iso.model.set_params(_rand_model([F, 20, 6, nout], seed=2).get_flat().astype(np.float32))      # (whatever a poisoned step left behind)
        pkg.run(iso, 1, 1)
        assert np.isfinite(iso.losses[-1]) and np.all(np.isfinite(pkg.chis(iso)))
    finally:
        iso.close()


def test_identical_samples_raise_a_domain_error(gpu_pkg):
    pkg = gpu_pkg
    A, F, N, K = 12, 40, 64, 3
    x1, y1 = O.synthetic_coords(A, 1, K, seed=3)
    xs = np.repeat(np.asarray(x1), N, axis=1)
    ys = np.repeat(np.asarray(y1)[:, :1, :], K, axis=1).repeat(N, axis=2)
    for nout in (1, 3):
        iso = _iso(pkg, xs, ys, A, F, nout)
        with pytest.raises(pkg.DomainError):
            pkg.run(iso, 1, 1)
        iso.close()


def test_coinciding_atoms_give_finite_results(gpu_pkg):
    """every atom at the origin: all distances 0, LayerNorm standardises a constant row (0 / eps), chi is the same for every sample --
    finite values, a DomainError from the target (constant chi), a finite (zero) coordinate gradient"""
    pkg = gpu_pkg
    A, F, N, K = 12, 40, 32, 2
    xs, ys = np.zeros((3 * A, N), np.float32), np.zeros((3 * A, K, N), np.float32)
    iso = _iso(pkg, xs, ys, A, F, 1)
    c = pkg.chis(iso)
    assert np.all(np.isfinite(c)) and np.ptp(c) == 0.0
    g = pkg.chicoords_pullback(iso, xs[:, :2], np.ones((1, 2)))
    assert np.all(g == 0.0)
    with pytest.raises(pkg.DomainError):
        pkg.run(iso, 1, 1)
    iso.close()


def test_single_sample_batches(gpu_pkg):
    """minibatch = 1 (the reference allows it: src/iso.jl:180-181): N optimizer steps of one row each, against the oracle"""
    pkg = gpu_pkg
    A, F, N, K = 12, 40, 24, 2
    xs, ys = O.synthetic_coords(A, N, K, seed=5)
    pairs = O.random_pairs(A, F, seed=1)
    om = _rand_model([F, 20, 6, 1], seed=2)
    odata = O.SimulationData(np.asarray(xs, np.float64), np.asarray(ys, np.float64), O.FeaturesPairs(pairs))
    iso = pkg.Iso(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), model=_mk_model(pkg, om), minibatch=1, opt=pkg.AdamRegularized())
    tgt = np.asarray(np.random.default_rng(0).uniform(size=(1, N)), np.float32)
    perm = np.random.default_rng(1).permutation(N)
    lo = O.train_batch(om, odata.features[0], tgt.astype(np.float64), O.setup(O.AdamRegularized(), om), 1, perm=perm)
    l = pkg.train_batch(iso, tgt, perm=perm)
    assert abs(l - lo) <= 2e-4 * abs(lo)
    np.testing.assert_allclose(iso.model.get_params(), om.get_flat(), atol=3e-5, rtol=3e-4)
    iso.close()
