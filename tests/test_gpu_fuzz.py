"""Randomized parity sweep of the C ABI against the fp64 oracle (scripts/fuzz_parity.py): shapes no parametrized test lists -- odd
N / K / F, ragged and oversized minibatches, 1 .. 10 outputs (shift-scale, device ISA, host ISA), one to four layers, both optimizers,
every forward mode, both featurize modes, identity / all-pairs / pair-list featurizers.  Per case: chi and K chi of the resident data,
then one epoch on the ENGINE's own target against the oracle's train_batch! (teacher-forced: an ill-conditioned simplex amplifies the
forward's 2e-7 beyond any fixed bound).  Round 5 ran 3 700 cases of this sweep.  The first 1 300 showed fourteen mismatches, every one
of them conditioning, none arithmetic: epochs that DIVERGE in the oracle too (one-layer models or un-normalised features at lr 1e-3:
parameters of 1e3 .. 1e11 within nine steps), LayerNorm over two nearly equal features, simplices with fewer points than vertices, a
Float64-exactly singular simplex that is only nearly singular behind a Float32 forward.  Those are now detected on the oracle's side and
skipped (about 7 % of the draws); the last 2 400 cases: no mismatch."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.gpu, pytest.mark.own_fwd_mode]


@pytest.mark.parametrize("seed", [11, 12])
def test_randomized_shapes_against_the_oracle(gpu_pkg, seed, monkeypatch):
    monkeypatch.delenv("ISK_FWD_MODE", raising=False)          # the sweep draws the forward mode per case
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "scripts", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    fails, skipped, failed = fz.sweep(150, seed, verbose=False)
    assert fails == 0, failed
    assert skipped < 30
