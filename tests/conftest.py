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
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: CPU test that takes more than a few seconds')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz')) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def relerr(a, b):
    """max|a-b| / max|b| (the tolerance measure of SURVEY.md 8d)."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    scale = np.max(np.abs(b))
    if scale == 0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b)) / scale)
