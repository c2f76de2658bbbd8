import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def built_lib():
    """The in-tree HIP library; built on demand (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    from edgegan_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.load_library()
