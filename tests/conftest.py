import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max |b| -- the 'relative to max' error used for all fp parity checks."""
    a, b = np.asarray(a), np.asarray(b)
    dt = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64   # complex spectra: modulus of the difference
    a, b = a.astype(dt), b.astype(dt)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
