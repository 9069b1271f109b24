import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/hades_oracle.c), built on demand.  Test infrastructure only."""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def hades_lib():
    """The product library through its C ABI (ctypes).  Built on demand with hipcc."""
    from hades252_amd import build, _lib
    build.build(verbose=False)
    return _lib.lib()


@pytest.fixture(scope="session")
def kat():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        return json.load(f)
