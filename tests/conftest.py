import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs an MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


@pytest.fixture(scope='session', autouse=True)
def _oracle_c_lib():
    """The oracle's C restatement is test infrastructure: (re)build it once per session."""
    import __graft_entry__ as g
    g.build_oracle()


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir('/root/reference/multipoint')
    skip_ref = pytest.mark.skip(reason='/root/reference not present (GPU box)')
    for item in items:
        if 'reference' in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(scope='session')
def oracle():
    from oracle import mp_oracle
    return mp_oracle


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
