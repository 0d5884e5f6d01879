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

# the clouds the training step really hands to the auction (bench.py c5_inputs): points sampled on K small primitives
# against a ground-truth cloud that fills the cube -- crowded, 500-1500 bidders per round
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for B, K, npp in ((64, 64, 32), (8, 16, 128)):
    params, gt = bench.synth_inputs(B, K, K * npp, 1234, torch.device('cuda'))
    kinds = vpn_amd.kinds_tensor([vpn_amd.SPHERE] * K, 'cuda')
    pred = vpn_amd.Sampling.sample_primitives(params, kinds, npp, seed=1234)
    d, a = emd(pred, gt, 0.005, 50); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        d, a = emd(pred, gt, 0.005, 50)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print('step clouds B=%d K=%d n=%d: %.3f ms  (%.1f us/sample)  EMD=%.5f' % (B, K, K * npp, dt * 1e3, dt * 1e6 / B, d.sqrt().mean().item()), flush=True)
