import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The C-ABI library, loaded through the product's own loader (fails loudly when it is not built)."""
    from iisan_amd import _lib
    return _lib.load()


@pytest.fixture(autouse=True)
def _dev_switches_back_at_their_defaults():
    """a test that leaves a development switch of the library (include/iisan_hip.h, DEV section) off its default silently
    re-routes every later test of the process.  After each test every switch must read its library default again; the fixture
    restores them so that one offender does not cascade, and fails the offender."""
    yield
    from iisan_amd import _lib
    if _lib._lib is None:              # the library was never loaded by this process: nothing to check
        return
    left = _lib.dev_state()
    if left:
        _lib.dev_reset()
        pytest.fail(f"the test left development switches off their library defaults: {left}")
