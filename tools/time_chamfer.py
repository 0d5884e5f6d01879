import sys, torch
sys.path.insert(0, '.')
import vpn_amd
from bench import synth_inputs
dev = torch.device('cuda')
B, K, n, M = 64, 32, 256, 2048
params, gt = synth_inputs(B, K, M, 1234, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
pts = vpn_amd.Sampling.sample_primitives(params, kinds, n, seed=1234)
for mode in sys.argv[1:]:
    for _ in range(3): vpn_amd.chamfer_nn(pts, gt, mode=mode)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): vpn_amd.chamfer_nn(pts, gt, mode=mode)
    b.record(); torch.cuda.synchronize()
    print('%-8s both directions: %.1f us' % (mode, a.elapsed_time(b) * 1e3 / 20))
