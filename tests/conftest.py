"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` tests: oracle vs analytic answers / golden vectors / torch autograd,
host logic, C-ABI export check.  `-m gpu` tests: HIP path vs the oracle, through the C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "own_fwd_mode: the test selects the forward arithmetic itself (not run once per mode)")


# forward arithmetic of the Koopman forward GEMM (isk_config.forward_mode): every GPU parity test runs in the f32-MFMA,
# the split-bf16 and the split-f16 mode (the default of bench.py); plain bf16 (mode 2) is reduced precision and has
# tests of its own.  ISK_TEST_FWD_MODES=3 (comma separated) restricts the sweep.
FWD_MODES = {0: "f32", 1: "bf16x3", 3: "f16x2"}


@pytest.fixture(params=[int(m) for m in os.environ.get("ISK_TEST_FWD_MODES", "0,1,3").split(",")], ids=lambda m: FWD_MODES.get(m, str(m)))
def fwd_mode(request, monkeypatch):
    if request.node.get_closest_marker("own_fwd_mode"):
        if request.param != int(os.environ.get("ISK_TEST_FWD_MODES", "0").split(",")[0]):
            pytest.skip("selects its own forward mode")
        monkeypatch.delenv("ISK_FWD_MODE", raising=False)
    else:
        monkeypatch.setenv("ISK_FWD_MODE", str(request.param))
    return request.param


@pytest.fixture(scope="session")
def pkg():
    """the product package (directory name `isokann.jl_amd` is not an identifier)"""
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    """package with the HIP library loaded and a device present"""
    lib = pkg.capi.load()          # raises loudly if the .so is missing
    n = pkg.capi.device_count()
    if n < 1:
        pytest.fail("gpu test selected but no HIP device is visible")
    return pkg


@pytest.fixture(scope="session")
def experimental_lib(gpu_pkg):
    """the tests of measured-and-rejected variants (persistent epoch, forward + tail as one launch, epoch-wide gather) need a
    library built with -DISK_EXPERIMENTAL (DEVFLAGS_ALL=-DISK_EXPERIMENTAL scripts/devbuild.sh devlib/libisokann_hip_exp.so,
    then ISK_LIB=devlib/libisokann_hip_exp.so pytest ...); the shipped library carries neither the options nor the kernels"""
    lib = gpu_pkg.capi.load()
    try:
        lib.isk_experimental_build
    except AttributeError:
        pytest.skip("shipped library: experimental variants are not built (-DISK_EXPERIMENTAL)")
    return lib
