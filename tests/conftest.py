import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'range_guard: tests/test_gpu_split_range.py - run with the range guard of the H = 256 Linear switched on')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


@pytest.fixture(scope='session')
def golden():
    return load_golden
