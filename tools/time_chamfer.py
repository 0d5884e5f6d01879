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
# how many queries the filter leaves to the fix-up kernel (counter at the head of each direction's list)
N = pts.shape[1]
ws = torch.empty((_lib.lib().vpn_chamfer_workspace(B, N, M) // 4,), dtype=torch.float32, device=dev)
d1 = torch.empty(B, N, device=dev); d2 = torch.empty(B, M, device=dev)
i1 = torch.empty(B, N, dtype=torch.int32, device=dev); i2 = torch.empty(B, M, dtype=torch.int32, device=dev)
_lib.call('vpn_chamfer_fwd_ws', _lib.ptr(pts), _lib.ptr(gt), B, N, M, _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2), _lib.ptr(i2),
          _lib.ptr(ws), ws.numel() * 4, 6, _lib.stream())
torch.cuda.synchronize()
pad = lambda n: (n + 63) & ~63
p4 = lambda n: (n + 3) & ~3
wi = ws.view(torch.int32)
SLOTS = 64
off1 = B * 16 * pad(M) + 3072 + B * SLOTS                # direction 1: targets p2 (M), queries p1 (N)
size1 = off1 + p4(B) + 4 * B * N + B * pad(M) + B * (pad(M) // 32) * 8   # + permutation + block boxes
off2 = size1 + B * 16 * pad(N) + 3072 + B * SLOTS
c1, c2 = wi[off1:off1 + B].cpu(), wi[off2:off2 + B].cpu()
print('undecided dir1 (8192 queries/sample): total %d max %d | dir2 (2048 queries/sample): total %d max %d' % (int(c1.sum()), int(c1.max()), int(c2.sum()), int(c2.max())))
print('dir2 per sample:', sorted(c2.tolist())[-10:], 'dir1:', sorted(c1.tolist())[-10:])
