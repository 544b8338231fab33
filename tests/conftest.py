import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_libs():
    from oracle import orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hip_libs():
    """The product library must already be built (in-tree .so); build it if hipcc is around."""
    from wgsparkl_amd import _ffi
    for dim in (2, 3):
        if not os.path.exists(_ffi.lib_path(dim)):
            _ffi.build()
    return _ffi
