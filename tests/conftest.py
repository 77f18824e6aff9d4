import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle_binding
    return oracle_binding.Oracle()


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def gpu(pkg):
    """One context for the whole GPU session (a single process, a single stream)."""
    so = pkg.lib_path()
    assert os.path.exists(so), "libqpgpu.so missing: the GPU tests never fall back to CPU code"
    g = pkg.QpGpu(0)
    yield g
    g.close()
