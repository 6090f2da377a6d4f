import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    config.addinivalue_line('markers', 'slow: minutes of CPU oracle at full size (still part of -m gpu)')


@pytest.fixture(scope='session')
def lib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    from unires_amd import _lib
    return _lib.load()


@pytest.fixture(scope='session')
def dev(lib):
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
