import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


@pytest.fixture(scope='session', params=['toy_a', 'toy_b'])
def golden(request):
    g = dict(np.load(os.path.join(GOLDEN, request.param + '.npz')))
    g['name'] = request.param
    g['path'] = os.path.join(GOLDEN, request.param)
    return g


@pytest.fixture(scope='session', autouse=True)
def _developer_tuning():
    """IGCN_TEST_TUNING="spmm_fold=0,topk_fast_mode=2": run the whole suite under non-default launch knobs (developer A/Bs of a
    kernel form that is not the default).  Unset: the library's defaults."""
    spec = os.environ.get('IGCN_TEST_TUNING', '')
    if spec:
        from igcn_cf_amd import _lib
        for item in spec.split(','):
            name, value = item.split('=')
            _lib.set_tuning(name.strip(), int(value))
    yield
