import sys, torch
sys.path.insert(0, '.')
import vpn_amd
from vpn_amd import _lib
from bench import synth_inputs
dev = torch.device('cuda')
B, K, H, W = 64, 32, 256, 256
params, gt = synth_inputs(B, K, 2048, 1234, dev)
p2, _ = synth_inputs(B, K, 2048, 4321, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
cam = torch.tensor([[1.0, 0.0, 0.0]], device=dev).expand(B, 3).contiguous()
with torch.no_grad():
    a2, d2 = vpn_amd.RasterFunction.apply(p2, kinds, cam, H, W, 0.05, 0.1, 2.0)
gs, gd = (a2 > 0.5).float(), d2.clone()
p = params.clone().requires_grad_(True)
def step():
    p.grad = None
    out = vpn_amd.RasterLossFunction.apply(p, kinds, cam, gs, gd, H, W, 0.05, 0.1, 2.0, False)
    out.sum().backward()
for _ in range(3): step()
with _lib.KernelProfile() as kp:
    for _ in range(20): step()
print({k: round(v[1] * 1e3, 1) for k, v in kp.summary().items() if 'raster' in k})
