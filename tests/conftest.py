import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    """ctypes handle on the CPU oracle (built on demand with gcc)."""
    from tests import _harness

    return _harness.oracle()


@pytest.fixture(scope="session")
def gpu():
    """The HIP engine bound through ctypes; fails loudly when there is no GPU
    or the extension is missing — GPU tests never fall back to a CPU path."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    from tests import _harness

    return _harness.engine()
