import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fixtures_lsd():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "lsd_fixtures.npz"))


@pytest.fixture(scope="session")
def built_lib():
    """Build (if stale) and load liblinefront.so.  Never substitutes anything for it."""
    from lineslam_amd import build as B
    from lineslam_amd import capi
    B.build()
    return capi.lib()
