"""ONE host process driving several engines (isk_dp_create_all / isk_dp_set_data_all / isk_dp_run_all): the single-process
multi-GPU model of SURVEY.md 8(b) -- the reference is one Julia process -- with a host thread per engine inside the call.
Build boxes have one GPU and real RCCL refuses two ranks on one device, so the engines share cuda:0 and the collectives go
through tests/mock_rccl (ISK_RCCL_LIB, ncclCommInitAll).  The collective library is loaded once per process, so each case runs
in a child process of its own."""
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.dp_checks import assert_params_match  # noqa: E402
from tests.test_gpu_dp2 import LR, _mk_iso, _problem  # noqa: E402
from tests.test_gpu_dp_native2 import mock_lib  # noqa: E402,F401  (fixture)

pytestmark = pytest.mark.gpu


def _child(_, mock, outdir, nout, minibatch, niter, n_eng, opt, inject, peer=0):
    os.environ["ISK_RCCL_LIB"] = mock
    os.environ["ISK_DP_PEER_REDUCE"] = str(peer)           # (the option's default for the engines this process creates)
    import importlib
    import __graft_entry__ as ge
    pkg = ge.load_package()
    dp = importlib.import_module("isokann_jl_amd.dp")
    xs, ys, pairs, widths, theta = _problem(nout)
    model = pkg.densenet(widths, layernorm=True)
    model.params = theta
    popt = pkg.AdamRegularized(LR, 1e-3) if opt == "adam" else pkg.NesterovRegularized(LR, 1e-3)
    tr = pkg.TransformShiftscale() if nout == 1 else pkg.TransformISA()
    grp = dp.IsoAll(pkg.SimulationData(xs, ys, pkg.FeaturesPairs(pairs)), [0] * n_eng, opt=popt, model=model, target=tr, minibatch=minibatch)
    nl = grp.n_local
    perms = [np.stack([np.random.default_rng(1000 * r + i).permutation(nl) for i in range(niter)]).astype(np.int32) for r in range(n_eng)]
    err = ""
    if inject:
        grp.set_option("debug_inject_rank_error", inject, rank=n_eng - 1)
    try:
        grp.run(niter, 1, perms=perms)
    except pkg.IskError as ex:
        err = f"{ex.status}|{ex}"
    out = dict(losses=np.asarray(grp.losses), chi=grp.chis(), err=err, perms=np.stack(perms), nl=nl, peer_active=grp.get_option("dp_peer_active"))
    for r in range(n_eng):
        out[f"params{r}"] = grp.get_params(r)
    if inject:
        grp.set_option("debug_inject_rank_error", 0)
        grp.run(1, 1, perms=[p[:1] for p in perms])
        out["losses_again"] = np.asarray(grp.losses)
    np.savez(os.path.join(outdir, "out.npz"), **out)
    grp.close()


def _run_child(mock, *args):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_child, args=(mock, d) + args, nprocs=1, join=True)
        return dict(np.load(os.path.join(d, "out.npz")))


@pytest.mark.parametrize("nout,minibatch,n_eng,opt", [(1, 64, 2, "nesterov"), (3, 64, 2, "adam"), (3, 0, 2, "nesterov"), (3, 48, 3, "nesterov")])
def test_one_process_several_engines_match_single_engine(gpu_pkg, mock_lib, nout, minibatch, n_eng, opt):   # noqa: F811
    pkg = gpu_pkg
    niter = 3
    o = _run_child(mock_lib, nout, minibatch, niter, n_eng, opt, 0)
    assert str(o["err"]) == ""
    for r in range(1, n_eng):
        np.testing.assert_array_equal(o["params0"], o[f"params{r}"])          # rank-ordered reduction: bit-identical replicas
    xs, ys, pairs, widths, theta = _problem(nout)
    N, nl = xs.shape[1], int(o["nl"])
    assert nl * n_eng == N
    bs = N if (minibatch == 0 or N < minibatch) else minibatch
    bl = nl if bs == N else bs // n_eng
    nb = 1 if bs == N else nl // bl
    gperms = []
    for it in range(niter):
        head = np.concatenate([np.concatenate([r * nl + o["perms"][r][it][b * bl:(b + 1) * bl] for r in range(n_eng)]) for b in range(nb)])
        gperms.append(np.concatenate([head, np.setdiff1d(np.arange(N), head)]))
    gb = bl * n_eng if bs != N else 0
    mk = lambda: _mk_iso(pkg, xs, ys, pairs, widths, theta, nout, gb, opt=opt)      # noqa: E731
    ref = mk()
    pkg.run(ref, niter, 1, perms=np.stack(gperms))
    np.testing.assert_allclose(o["losses"], ref.losses, rtol=2e-4)
    want, chi = ref.model.get_params(), pkg.chis(ref)
    ref.close()
    assert_params_match(pkg, o["params0"], want, opt, mk, gperms[0][:gb if gb else N], LR, niter * nb)
    if opt == "nesterov":
        np.testing.assert_allclose(o["chi"], chi, atol=2e-5)                 # chis() of the group = the engines' slabs in rank order


def test_one_process_rank_error_stops_every_engine_in_the_same_step(gpu_pkg, mock_lib):   # noqa: F811
    """the last engine flags a failure at its 2nd optimizer step: isk_dp_run_all returns THAT rank's error (not the ISK_EPEER it
    causes on the others), the replicas stay bit-identical, the group runs on afterwards"""
    pkg = gpu_pkg
    o = _run_child(mock_lib, 3, 32, 2, 2, "nesterov", 2)
    err = str(o["err"])
    assert err.startswith(f"{pkg.capi.EHIP}|") and "rank 1" in err and "injected rank error" in err, err
    np.testing.assert_array_equal(o["params0"], o["params1"])
    assert len(o["losses"]) == 0 and len(o["losses_again"]) == 1 and np.isfinite(o["losses_again"][0])


@pytest.mark.parametrize("nout,minibatch,n_eng,opt", [(3, 64, 2, "adam"), (3, 48, 3, "nesterov"), (1, 0, 2, "nesterov")])
def test_peer_reduce_equals_the_all_reduce_bit_for_bit(gpu_pkg, mock_lib, nout, minibatch, n_eng, opt):   # noqa: F811
    """option dp_peer_reduce: no gradient all-reduce -- every engine's optimizer step reads the other engines' gradient slabs
    directly (peer-mapped; here the engines share one device) and sums them in rank order.  The mock collective library reduces in
    rank order too, so the two runs must agree in every bit: losses, parameters of every replica, chi."""
    a = _run_child(mock_lib, nout, minibatch, 3, n_eng, opt, 0, 1)
    b = _run_child(mock_lib, nout, minibatch, 3, n_eng, opt, 0, 0)
    assert str(a["err"]) == "" and str(b["err"]) == ""
    assert int(a["peer_active"]) == 1 and int(b["peer_active"]) == 0
    np.testing.assert_array_equal(a["losses"], b["losses"])
    for r in range(n_eng):
        np.testing.assert_array_equal(a[f"params{r}"], b["params0"])
    np.testing.assert_array_equal(a["chi"], b["chi"])


def test_peer_reduce_rank_error_stops_every_engine_in_the_same_step(gpu_pkg, mock_lib):   # noqa: F811
    """the error flag of a replica travels in the slot behind the batch loss of its gradient slab: the peer sum carries it to every
    engine in the same step, exactly as the all-reduce does"""
    pkg = gpu_pkg
    o = _run_child(mock_lib, 3, 32, 2, 2, "nesterov", 2, 1)
    err = str(o["err"])
    assert err.startswith(f"{pkg.capi.EHIP}|") and "rank 1" in err and "injected rank error" in err, err
    np.testing.assert_array_equal(o["params0"], o["params1"])
    assert len(o["losses"]) == 0 and len(o["losses_again"]) == 1 and np.isfinite(o["losses_again"][0])
