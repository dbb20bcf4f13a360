"""N beyond what the parametrized tests use (70 000 ... 1 200 000 start points, tiny features): the ISA chain's partial-record staging
leaves its LDS fast path (more than 2048 partial doubles: N > 230 000 at three outputs), minibatch plans with hundreds of steps, the
shift-scale reductions over a million rows -- target and one epoch of train_batch! against the oracle (scripts/probe_big_n.py)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.gpu]


def test_large_n_target_and_epoch_match_the_oracle(gpu_pkg, capsys):
    spec = importlib.util.spec_from_file_location("probe_big_n", os.path.join(ROOT, "scripts", "probe_big_n.py"))
    pb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pb)
    fails = pb.main()
    out = capsys.readouterr().out
    assert fails == 0 and out.count("PASS") == 5, out
