import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def native():
    """The ctypes binding (loads without a GPU; build() first if missing)."""
    lib = os.path.join(ROOT, "minbpe_amd", "lib", "libbpe_hip.so")
    if not os.path.exists(lib):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    from minbpe_amd import _native
    return _native


@pytest.fixture(scope="session")
def engine(native):
    eng = native.Engine(0)
    yield eng
    eng.close()
