import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cse168-raytracer_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/): the checker only -- never the thing under test or measured."""
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def miro():
    """The product binding; raises loudly when the HIP library has not been built."""
    import miro_amd
    miro_amd.load_library()
    return miro_amd


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
