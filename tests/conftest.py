import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    from strkit_amd import _lib
    return _lib.default_context(0)


@pytest.fixture
def fresh_ctx():
    """A context of its own: a shared one carries what earlier batches taught it (a band cool-down after noisy reads, grid
    history), which a test that asserts on the band's statistics must not depend on."""
    from strkit_amd import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
