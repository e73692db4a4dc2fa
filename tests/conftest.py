import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fie():
    """The product's HIP context; raises (never falls back) when the library or the GPU is missing."""
    import fie_amd  # noqa: F401
    from fie_amd import hip
    return hip.context(0)
