import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the HIP library and the oracle exist (hipcc cross-compiles without a GPU)."""
    from cmc_fluid_solver_amd import build as b
    b.build()
    from oracle import oracle as O
    O.build()
    return True
