"""Shapes at the edges of what isk_create accepts (scripts/shape_probe.py): feature rows of 4 ... 10000 on the identity featurizer, 2 ... 100
atoms on the all-pairs one, pair lists on 682 / 683 / 2000 / 3000 atoms and with 2049 / 4000 pairs, hidden widths around every
kernel-selection boundary (1, 160 / 161, a 16- / 17-wide tail, 512 / 513, eight layers, 64 outputs), N and K down to 2 and 1 -- in
every forward arithmetic: chi and one epoch of train_batch! against the oracle, the coordinate pullback finite.  Written in round 5 after
such a probe found that hidden widths 2560 ... 4096 were accepted and failed at the first launch."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.gpu, pytest.mark.own_fwd_mode]


def test_edge_shapes_against_the_oracle(gpu_pkg, monkeypatch, capsys):
    monkeypatch.delenv("ISK_FWD_MODE", raising=False)          # the probe runs every forward mode itself
    spec = importlib.util.spec_from_file_location("shape_probe", os.path.join(ROOT, "scripts", "shape_probe.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    rc = sp.main()
    out = capsys.readouterr().out
    assert rc == 0, "\n".join(l for l in out.splitlines() if l.startswith("FAIL"))
    assert out.count("PASS") >= 140
