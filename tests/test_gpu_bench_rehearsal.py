"""bench.py's multi-rank branch, rehearsed before an 8-GPU node sees it: `python -m torch.distributed.run --nproc-per-node 2
bench.py --gpus 2` on ONE GPU -- torch.distributed on gloo (ISK_BENCH_DIST_BACKEND), the engine's collectives through
tests/mock_rccl (ISK_RCCL_LIB; real RCCL refuses two ranks on one device).  Checks what the driver's SCALE run relies on: exactly
one JSON line on stdout, n_gpus = 2, strong scaling of C4 (global N and global minibatch sharded), replicas consistent, both
sub-records present, the watchdog silent, exit code 0 -- and that the line says it is a rehearsal, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.test_gpu_dp2 import _free_port  # noqa: E402
from tests.test_gpu_dp_native2 import mock_lib  # noqa: E402,F401  (fixture)

pytestmark = [pytest.mark.gpu, pytest.mark.own_fwd_mode]


def _launch(mock, extra, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("ISK_FWD_MODE", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(ISK_BENCH_DIST_BACKEND="gloo", ISK_RCCL_LIB=mock, ISK_BENCH_WATCHDOG_S="120", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"] + extra
    return subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=timeout)


def test_two_rank_bench_line_on_one_gpu(gpu_pkg, mock_lib):  # noqa: F811
    r = _launch(mock_lib, [])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line; nothing else reaches stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 6
    assert d["dp_replicas_consistent"] is True
    cfg = d["config"]
    assert cfg["N_global"] == 10000 and cfg["N_per_gpu"] == 5000 and cfg["minibatch_global"] == 1000 and cfg["minibatch_per_gpu"] == 500
    assert d["value"] == pytest.approx(cfg["N_global"] * cfg["K"] / (d["ms_per_step"] * 1e-3), rel=1e-6)   # whole-job aggregate
    sub = d["sub_records"]
    assert set(sub) == {"weak", "mb0"}, sub
    assert "error" not in sub["weak"] and sub["weak"]["scaling"] == "weak" and sub["weak"]["N_global"] == 20000
    assert "error" not in sub["mb0"] and sub["mb0"]["minibatch_global"] == 0 and sub["mb0"]["N_global"] == 10000
    assert "watchdog_hang" not in d and "watchdog" not in r.stderr.lower().replace("isk_bench_watchdog_s", "")
    assert d["rehearsal"]["backend"] == "gloo" and d["rehearsal"]["ranks"] == 2
    assert d["roofline"]["training_chain"]["allreduce_ms_per_epoch"] is not None        # the collectives were timed on the engine's stream
    assert "native data-parallel step failed" not in r.stderr                            # the engine-driven run, not the host-driven fallback


def test_two_rank_bench_weak_scaling_and_host_driven_collectives(gpu_pkg, mock_lib):  # noqa: F811
    """--scaling weak (a full C3 per rank: the sub-record `weak` is then the headline and is not repeated) and ISK_DP_HOST=1
    (dp.py issues the collectives itself through torch.distributed: the fallback of the first-step safety valve)"""
    r = _launch(mock_lib, ["--scaling", "weak", "--no-sub-records"], {"ISK_DP_HOST": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["N_global"] == 20000 and d["dp_replicas_consistent"] is True
    assert d["sub_records"] is None
