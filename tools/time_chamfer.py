"""Chamfer scan at the C3 workload: per-kernel times (library launch profiler), undecided counts."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd
from vpn_amd import _lib
from bench import synth_inputs
dev = torch.device('cuda')
B, K, n, M = 64, 32, 256, 2048
params, gt = synth_inputs(B, K, M, 1234, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
pts = vpn_amd.Sampling.sample_primitives(params, kinds, n, seed=1234)
for mode in (sys.argv[1:] or ["mfma16"]):
    for _ in range(3): vpn_amd.chamfer_nn(pts, gt, mode=mode)
    with _lib.KernelProfile() as kp:
        for _ in range(20): vpn_amd.chamfer_nn(pts, gt, mode=mode)
    r = {k: round(v[1] * 1e3, 1) for k, v in kp.summary().items()}
    print('%-8s %s  sum %.1f us' % (mode, r, sum(r.values())))
d1, i1, d2, i2 = vpn_amd.chamfer_nn(pts, gt, mode="mfma16")
e1, j1, e2, j2 = vpn_amd.chamfer_nn(pts, gt, mode='brute')
print('mfma == brute:', bool(torch.equal(d1, e1) and torch.equal(i1, j1) and torch.equal(d2, e2) and torch.equal(i2, j2)))
import ctypes
L = _lib.lib()
if hasattr(L, 'vpn_debug_read'):
    buf = (ctypes.c_ulonglong * 8)()
    L.vpn_debug_read(buf)
    print('debug counters:', list(buf))
