import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # tests/test_dist_gpu.py forks its rank processes from a fork SERVER; the server itself is spawned (fork + exec) here,
    # before any test has initialised the GPU: a process that has touched the GPU must never exec (the pool forbids it)
    import multiprocessing
    import multiprocessing.forkserver as fs
    try:
        multiprocessing.get_context('forkserver')
        fs.ensure_running()
    except Exception:              # no fork server on this platform: the test that needs it fails by itself
        pass


def load_golden(name):
    """Golden vectors captured from the reference's own leaf files by
    oracle/make_golden.py (plain .npz, allow_pickle=False)."""
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: torch.from_numpy(np.array(z[k])) for k in z.files}


@pytest.fixture
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max |b|  (norm-wise relative error, the 1e-4 fp32 bar of north_star)."""
    a, b = a.double(), b.double()
    den = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / den)


def elem_rel_err(a, b, floor=1e-2):
    """Element-wise relative error with an absolute floor: max |a-b| / (|b| + floor * max|b|).  Unlike `rel_err`
    it holds small components to (nearly) their own scale: a component 100x below the largest is still checked
    to twice the stated tolerance of its own size."""
    a, b = a.double(), b.double()
    den = b.abs() + floor * b.abs().max().clamp_min(1e-30)
    return float(((a - b).abs() / den).max())
