"""The boundary without Python in between: tests/c_abi/run_c1.c, a plain C99 program, links libisokann_hip.so and drives
configuration C1 through isk_config_default -> isk_create -> isk_set_params / isk_set_data -> isk_target -> isk_run -> isk_chis,
the sequence a Julia host's ccalls make (julia/IsokannHIP.jl; reference: src/iso.jl:17-43,72-94,203, src/isotarget.jl:10-12),
and compares with the committed golden vectors (c1_golden.h = tests/golden/c1_shiftscale_nesterov.npz, make_c1_header.py)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CDIR = os.path.join(HERE, "c_abi")
LIBDIR = os.path.join(ROOT, "isokann.jl_amd", "lib")


def _build(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    exe = str(tmp_path / "run_c1")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + CDIR,
                    os.path.join(CDIR, "run_c1.c"), "-L" + LIBDIR, "-lisokann_hip", "-lm", "-Wl,-rpath," + LIBDIR,
                    "-Wl,-rpath-link," + os.path.join(rocm, "lib"), "-o", exe], check=True)
    return exe


def test_golden_header_is_current():
    """c1_golden.h is the committed npz, value for value"""
    g = np.load(os.path.join(HERE, "golden", "c1_shiftscale_nesterov.npz"))
    txt = open(os.path.join(CDIR, "c1_golden.h")).read()

    def values(name):
        body = txt[txt.index(name + "["):]
        body = body[body.index("{") + 1:body.index("}")]
        return np.array([float(t.strip().rstrip("f")) for t in body.replace("\n", " ").split(",") if t.strip()])
    np.testing.assert_array_equal(values("c1_xs").astype(np.float32), g["xs"].reshape(-1, order="F").astype(np.float32))
    np.testing.assert_array_equal(values("c1_ys").astype(np.float32), g["ys"].reshape(-1, order="F").astype(np.float32))
    np.testing.assert_array_equal(values("c1_params0").astype(np.float32), g["params0"].astype(np.float32))
    np.testing.assert_array_equal(values("c1_perms"), g["perms"].reshape(-1))
    for k in ("losses", "params_final"):
        np.testing.assert_array_equal(values("c1_" + k), g[k].reshape(-1, order="F"))
    for k in ("target0", "chis", "koopman"):
        np.testing.assert_array_equal(values("c1_" + k), g[k].reshape(-1, order="F"))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_driver_compiles_as_c99_and_links_against_the_library(tmp_path):
    """strict C99, every warning an error: the header and the driver use nothing outside the C ABI.  Without a device the program
    must stop at isk_create with the library's own message (the product path has no CPU fallback)"""
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    import __graft_entry__ as ge
    if ge.load_package().capi.device_count() < 1:
        assert r.returncode == 2 and "isk_create" in r.stderr, (r.returncode, r.stderr)
    else:
        assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.own_fwd_mode
@pytest.mark.parametrize("mode", [0, 1, 3])
def test_c1_through_the_c_abi_from_plain_c(gpu_pkg, tmp_path, mode):
    exe = _build(tmp_path)
    env = {k: v for k, v in os.environ.items() if k != "ISK_FWD_MODE"}
    r = subprocess.run([exe, str(mode)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C1 through the C ABI: ok" in r.stdout and "MISMATCH" not in r.stdout
