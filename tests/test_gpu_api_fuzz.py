"""Random SEQUENCES of API calls against the oracle replaying the same sequence (scripts/api_fuzz.py): run, chis, koopman, isotarget,
train_batch, batch_grads, set_params, an optimizer-state round trip, a changed minibatch, chicoords, data growth (drop start points at
the front, append new ones: addcoords! / the cutoff window) and a save / load through a file into a NEW engine, in random order on random small
problems -- both featurizers, one to three hidden layers, the fused and the forced-layered path, every Float32-equivalent forward
arithmetic, per-iteration and cached featurization, both optimizers.  What it is after is the engine's cached derived state (weight
images and their dirty flags, feature buffers written ahead on the side stream, epoch graphs, the device copy of beta^t).  Round 5:
3 850 sequences, no mismatch once the oracle-side check for an UNDETERMINED ISA target was in place (an ill-conditioned simplex, or
two permutations of fixperm at the same cost to rounding -- K chi of a deep sigmoid network at initialisation lies almost on a line)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.gpu, pytest.mark.own_fwd_mode]


@pytest.mark.parametrize("seed", [21, 22])
def test_random_api_sequences_against_the_oracle(gpu_pkg, seed, monkeypatch):
    monkeypatch.delenv("ISK_FWD_MODE", raising=False)          # the sweep draws the forward mode per sequence
    monkeypatch.delenv("ISK_FORCE_LAYERED", raising=False)
    spec = importlib.util.spec_from_file_location("api_fuzz", os.path.join(ROOT, "scripts", "api_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    fails, skipped, failed = fz.sweep(150, seed, verbose=False)
    assert fails == 0, failed
    assert skipped < 40


def test_random_api_sequences_bf16_paths(gpu_pkg, monkeypatch):
    """the same on the plain-bf16 forward (forward_mode 2: the fused wide kernel on the layered path, the one-plane kernel on the fused
    one) with and without the bf16 training GEMMs: forward / one-epoch operations only, bf16 tolerances (4e-2 on chi, 6e-2 on a loss) --
    an order of magnitude below what a stale weight image costs after set_params"""
    monkeypatch.delenv("ISK_FWD_MODE", raising=False)
    monkeypatch.delenv("ISK_FORCE_LAYERED", raising=False)
    spec = importlib.util.spec_from_file_location("api_fuzz", os.path.join(ROOT, "scripts", "api_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    fails, skipped, failed = fz.sweep(150, 23, verbose=False, bf16=True)
    assert fails == 0, failed
