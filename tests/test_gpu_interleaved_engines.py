"""Several engines alive in one process on one GPU, their calls interleaved: every engine must produce exactly (bit for bit) what it
produces alone -- nothing of one engine's state may live in a static of the library (function attributes, staging buffers, streams,
graphs keyed by buffer, option defaults)."""
import numpy as np
import pytest

from oracle import isokann_oracle as O
from tests.test_gpu_parity import _mk_model, _rand_model

pytestmark = [pytest.mark.gpu]


def _mk(pkg, spec):
    rng = np.random.default_rng(spec["seed"])
    N, K = spec["N"], spec["K"]
    if spec["kind"] == "pairs":
        xs, ys = O.synthetic_coords(spec["A"], N, K, seed=spec["seed"])
        feat = pkg.FeaturesPairs(O.random_pairs(spec["A"], spec["widths"][0], seed=spec["seed"] + 1))
    else:
        xs, ys = rng.normal(size=(spec["widths"][0], N)), rng.normal(size=(spec["widths"][0], K, N))
        feat = pkg.identity
    om = _rand_model(spec["widths"], seed=spec["seed"] + 2)
    nout = spec["widths"][-1]
    iso = pkg.Iso(pkg.SimulationData(xs, ys, feat), model=_mk_model(pkg, om), minibatch=spec["mb"], target=pkg.TransformISA() if nout > 1 else None,
                  opt=pkg.AdamRegularized() if spec["opt"] == "adam" else pkg.NesterovRegularized(), forward_mode=spec["mode"],
                  featurize_every_iteration=spec["every"], force_layered=spec.get("layered", False), options=spec.get("options"))
    perms = np.stack([np.random.default_rng(spec["seed"] + 10 + i).permutation(N) for i in range(4)]).astype(np.int32)
    return iso, perms


SPECS = [
    dict(kind="pairs", A=30, widths=[120, 40, 6, 3], N=384, K=3, mb=64, opt="adam", mode=3, every=True, seed=1),
    dict(kind="identity", widths=[48, 300, 60, 1], N=257, K=2, mb=0, opt="nesterov", mode=1, every=False, seed=2, layered=True),
    dict(kind="pairs", A=22, widths=[231, 38, 6, 1], N=500, K=4, mb=100, opt="nesterov", mode=0, every=True, seed=3),
    dict(kind="pairs", A=40, widths=[200, 256, 256, 1], N=320, K=2, mb=64, opt="adam", mode=2, every=True, seed=4, layered=True, options={"train_bf16": 1}),
]


def test_interleaved_engines_equal_their_solo_runs(gpu_pkg):
    pkg = gpu_pkg
    solo = []
    for spec in SPECS:
        iso, perms = _mk(pkg, spec)
        out = []
        for it in range(4):
            pkg.run(iso, 1, 1, perms=perms[it:it + 1])
            out.append((iso.losses[-1], pkg.chis(iso).copy()))
        solo.append((out, iso.model.get_params().copy()))
        iso.close()
    live = [_mk(pkg, spec) for spec in SPECS]
    got = [[] for _ in SPECS]
    order = [0, 1, 2, 3, 3, 1, 0, 2, 2, 0, 3, 1, 1, 2, 3, 0]                  # four iterations each, interleaved
    nxt = [0] * len(SPECS)
    for e in order:
        iso, perms = live[e]
        it = nxt[e]
        pkg.run(iso, 1, 1, perms=perms[it:it + 1])
        got[e].append((iso.losses[-1], pkg.chis(iso).copy()))
        nxt[e] += 1
    for e, (iso, _) in enumerate(live):
        for it in range(4):
            assert got[e][it][0] == solo[e][0][it][0], (e, it, got[e][it][0], solo[e][0][it][0])
            np.testing.assert_array_equal(got[e][it][1], solo[e][0][it][1])
        np.testing.assert_array_equal(iso.model.get_params(), solo[e][1])
        iso.close()


def test_engines_on_concurrent_host_threads_equal_their_solo_runs(gpu_pkg):
    """the ABI is re-entrant per handle: one host thread per engine, running at the same time (ctypes releases the
    GIL around every call), must change nothing either"""
    import threading
    pkg = gpu_pkg
    solo = []
    for spec in SPECS:
        iso, perms = _mk(pkg, spec)
        pkg.run(iso, 4, 1, perms=perms)
        solo.append((list(iso.losses), pkg.chis(iso).copy(), iso.model.get_params().copy()))
        iso.close()
    live = [_mk(pkg, spec) for spec in SPECS]
    errs = []

    def work(e):
        try:
            iso, perms = live[e]
            for it in range(4):
                pkg.run(iso, 1, 1, perms=perms[it:it + 1])
                pkg.chis(iso)
        except Exception as ex:  # noqa: BLE001
            errs.append((e, repr(ex)))

    threads = [threading.Thread(target=work, args=(e,)) for e in range(len(SPECS))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for e, (iso, _) in enumerate(live):
        assert list(iso.losses) == solo[e][0], (e, iso.losses, solo[e][0])
        np.testing.assert_array_equal(pkg.chis(iso), solo[e][1])
        np.testing.assert_array_equal(iso.model.get_params(), solo[e][2])
        iso.close()


def test_engines_release_their_device_memory(gpu_pkg):
    """isk_destroy gives everything back: 60 engines of the four kinds above created, run and destroyed in turn leave the device's free
    memory where it was (to within an allocator granule)"""
    import torch
    pkg = gpu_pkg

    def cycle(n):
        for i in range(n):
            iso, perms = _mk(pkg, SPECS[i % len(SPECS)])
            pkg.run(iso, 1, 1, perms=perms[:1])
            pkg.chicoords_pullback(iso, iso.data.coords[0][:, :2], np.ones((iso.nout, 2)))
            iso.close()

    cycle(8)                                            # (first use: code objects, the runtime's own pools)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(60)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0 - free1) / 2 ** 20
