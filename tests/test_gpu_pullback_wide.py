"""d chi / d coords (the counterpart of sqpairdist_bwd_kernel! + the MLP pullback, src/utils/pairdists.jl:153-196) beyond the shapes the
register-resident featurizer covers -- more than 2048 features, more than 682 atoms: the generic featurizer keeps the raw distances and
the LayerNorm statistics too (round 5; until then such shapes were refused with ISK_EUNSUPPORTED).  Against torch's fp64 autograd of
the same network on the float32 coordinates, 2e-5 of the largest gradient entry."""
import numpy as np
import pytest

from oracle import isokann_oracle as O
from tests.test_gpu_round4 import _model

pytestmark = [pytest.mark.gpu]


@pytest.mark.parametrize("A,F,widths_tail,layered", [(200, 3000, [100, 10, 2], False), (120, 2500, [300, 40, 2], True), (0, 2100, [24, 4, 1], False),
                                                       (800, 1000, [60, 8, 3], False), (700, 4100, [48, 6, 2], False)])
def test_pullback_beyond_2048_features(gpu_pkg, A, F, widths_tail, layered):
    import torch
    pkg = gpu_pkg
    nout, ncol = widths_tail[-1], 6
    om, m = _model(pkg, [F] + widths_tail, seed=21)
    rng = np.random.default_rng(3)
    if A:
        xs, ys = O.synthetic_coords(A, 16, 2, seed=5)
        pairs = O.random_pairs(A, F, seed=0)
        data = pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs))
    else:
        xs, ys = rng.normal(size=(F, 16)), rng.normal(size=(F, 2, 16))
        pairs = None
        data = pkg.SimulationData(xs, ys, pkg.identity)
    iso = pkg.Iso(data, model=m, target=pkg.TransformISA() if nout > 1 else None, minibatch=0)
    cot = rng.normal(size=(nout, ncol))
    got = pkg.chicoords_pullback(iso, xs[:, :ncol], cot)
    x32 = np.asarray(xs[:, :ncol], np.float32).astype(np.float64)
    want = np.zeros_like(x32)
    for n in range(ncol):
        xt = torch.tensor(x32[:, n], requires_grad=True)
        if A:
            P = torch.tensor(np.asarray(pairs))
            c = xt.reshape(-1, 3)
            a = torch.sqrt(((c[P[:, 0]] - c[P[:, 1]]) ** 2).sum(dim=1))
        else:
            a = xt
        mu = a.mean()
        var = ((a - mu) ** 2).mean()
        a = (a - mu) / torch.sqrt(var + om.ln_eps ** 2) * torch.tensor(om.ln_scale) + torch.tensor(om.ln_bias)
        for W, b, act in zip(om.W, om.b, om.acts()):
            z = torch.tensor(W) @ a + torch.tensor(b)
            a = torch.sigmoid(z) if act == "sigmoid" else z
        (a * torch.tensor(cot[:, n])).sum().backward()
        want[:, n] = xt.grad.numpy()
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, atol=2e-5 * scale, rtol=2e-4)
    iso.close()
