import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)')


def _gpu_available():
    try:
        from qmps_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need libqmps_hip.so and a gfx950 device: on a box without one (and unless they were asked for
    with `-m gpu`) they are skipped instead of failing with QMPS_ERR_NO_DEVICE."""
    if 'gpu' in (config.getoption('-m') or '') and 'not gpu' not in (config.getoption('-m') or ''):
        return                                   # explicitly requested: fail loudly if the device is missing
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason='no gfx950 device / libqmps_hip.so on this box')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


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
