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
