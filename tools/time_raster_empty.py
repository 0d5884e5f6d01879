"""raster_total_kernel's fixed cost per tile: the C3 shape with every primitive moved out of the frustum (no visible
pairs) against the normal scene."""
import sys, torch
sys.path.insert(0, '.')
import vpn_amd
from vpn_amd import _lib
from bench import synth_inputs
dev = torch.device('cuda')
B, K, H = 64, 32, 256
W = H
params, _ = synth_inputs(B, K, 8, 1234, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
cam = torch.tensor([[1.0, 0.0, 0.0]], device=dev).expand(B, 3).contiguous()
gs = torch.zeros(B, H, W, device=dev); gd = torch.full((B, H, W), 2.0, device=dev)
one = torch.ones((), device=dev)
for name, shift in (('normal', 0.0), ('all primitives off screen', 50.0), ('behind camera', -50.0)):
    p = params.clone()
    p[..., 8] += shift                      # translate in y: far above the frustum
    p.requires_grad_(True)
    def step():
        p.grad = None
        out = vpn_amd.RasterTotalFunction.apply(p, kinds, cam, gs, gd, H, W, 0.05, 0.1, 2.0, False, 1.0, 1.0)
        out[2].backward(one)
    for _ in range(3): step()
    with _lib.KernelProfile() as kp:
        for _ in range(20): step()
    print(name, {k: round(v[1] * 1e3, 1) for k, v in kp.summary().items()})
