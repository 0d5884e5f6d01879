"""Time the auction kernel: python tools/time_emd.py [B n eps iters]..."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd
from vpn_amd import _lib
emd = vpn_amd.modules.loss.EarthMoverDistanceLoss()
cases = [(64, 2048, 0.005, 50), (8, 2048, 0.005, 50), (256, 2048, 0.005, 50), (20, 8192, 0.05, 3000)]
for B, n, eps, iters in cases:
    g = torch.Generator().manual_seed(1)
    x1 = torch.rand(B, n, 3, generator=g).cuda(); x2 = torch.rand(B, n, 3, generator=g).cuda()
    d, a = emd(x1, x2, eps, iters); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        d, a = emd(x1, x2, eps, iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    uniq = sum(a[b].unique().numel() for b in range(B)) / (B * n)
    print('B=%d n=%d eps=%g iters=%d: %.3f ms  (%.1f us/sample)  EMD=%.5f  unique=%.4f' % (B, n, eps, iters, dt * 1e3, dt * 1e6 / B, d.sqrt().mean().item(), uniq), flush=True)
