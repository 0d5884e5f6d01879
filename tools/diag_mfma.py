import sys, ctypes, torch
sys.path.insert(0, '.')
import vpn_amd
from vpn_amd import _lib
from bench import synth_inputs
dev = torch.device('cuda')
B, K, n, M = 64, 32, 256, 2048
params, gt = synth_inputs(B, K, M, 1234, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
pts = vpn_amd.Sampling.sample_primitives(params, kinds, n, seed=1234)
L = _lib.lib(); buf = (ctypes.c_ulonglong * 8)()
def rd():
    torch.cuda.synchronize(); L.vpn_debug_read(buf); return list(buf)
a = rd(); vpn_amd.chamfer_nn(pts, gt, mode='mfma'); b = rd()
print('ambiguity rescans (both directions): %d of %d queries' % (b[6] - a[6], B * (K * n + M)))
