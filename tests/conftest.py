import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'qmps_golden.npz'))


@pytest.fixture(scope='session')
def c_oracle():
    from oracle import c_oracle as C
    C.build()
    return C


_ENGINES = {}


@pytest.fixture(scope='session')
def engine_factory():
    """Engines are cached per (D, capacity) for the whole session: hipMalloc once."""
    from qmps_amd import EnergyEngine

    def get(D, max_batch=1 << 16):
        key = (D, max_batch)
        if key not in _ENGINES:
            _ENGINES[key] = EnergyEngine(D, max_batch)
        return _ENGINES[key]

    yield get
    for e in _ENGINES.values():
        e.close()
    _ENGINES.clear()
