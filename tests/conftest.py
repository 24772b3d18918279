import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def h2():
    """the product package (loads libh2mi.so; raises if it is not built)."""
    import _load_pkg

    return _load_pkg.load()


@pytest.fixture(scope="session")
def gpu(h2):
    """initialised device context; GPU tests fail (not skip) if the library cannot reach a GPU."""
    import torch  # noqa: F401  (import first so one HIP runtime is shared)

    h2.init(0)
    return h2
