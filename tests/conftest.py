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


def decidable_depth_gt(oracle, params, kinds, cam, gt_dep, H, W, sigma=0.05, gamma=0.1, z_far=2.0, chunk=2):
    """The L1 depth loss differentiates through sign(D - gt): where the predicted depth lies within fp32 rounding
    noise of the GT that sign is undecidable, and ONE flipped pixel moves the gradient of a large image by ~1e-4..1e-3
    relative (Finding 6).  Returns gt_dep with those pixels (|D - gt| < 1e-5, D from the oracle) moved by
    1e-3, so that a parity test compares implementations, not coin flips."""
    with torch.no_grad():
        d = torch.cat([oracle.raster(params[b:b + chunk], kinds, cam[b:b + chunk], H, W, sigma, gamma, z_far)[1]
                       for b in range(0, params.shape[0], chunk)])
    diff = d - gt_dep.reshape(d.shape)
    near = (diff != 0) & (diff.abs() < 1e-5)
    moved = gt_dep.reshape(d.shape) - torch.where(diff >= 0, torch.full_like(diff, 1e-3), torch.full_like(diff, -1e-3))
    return torch.where(near, moved, gt_dep.reshape(d.shape)).reshape(gt_dep.shape)
