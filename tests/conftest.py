import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _fresh_build():
    """never test a stale libhk.so / oracle: rebuild both when any source is newer (no-op otherwise)"""
    import __graft_entry__ as ge
    ge.build()
