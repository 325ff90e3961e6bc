import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run under gpurun)')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope='session')
def golden():
    return load_golden


def unpack_mask(packed, shape):
    n = int(np.prod(shape))
    bits = np.unpackbits(packed.numpy())[:n].reshape(shape)
    return torch.from_numpy(bits.astype(np.bool_))
