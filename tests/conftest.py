import os
import sys

# The CPU oracle (torch ops, OpenMP) spin-waits at its barriers by default: with as many threads as CPUs, ONE CPU the
# sandbox does not actually schedule (steal time / a parked vCPU -- observed in this container after the process-spawning
# tests: two of eight threads share a core, the other six spin) turns a 10-second test into a 20-minute one.  Passive waits
# and two CPUs of headroom keep the suite at a few minutes whatever the host does.  (Set before torch is imported.)
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
os.environ.setdefault('GOMP_SPINCOUNT', '1000')

import numpy as np
import pytest
import torch

torch.set_num_threads(max(1, min(6, (os.cpu_count() or 8) - 2)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def load_golden():
    return golden
