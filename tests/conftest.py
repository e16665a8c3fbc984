import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def jtk_lib():
    """libjtk_lc.so; built in-tree with hipcc if missing (cross-compiles without a GPU)."""
    from jtk_amd import build, ffi
    build.build()
    return ffi.lib()
