import sys, torch
sys.path.insert(0, '.')
import vpn_amd
from oracle import vpn_oracle as O
from bench import synth_inputs
dev = torch.device('cuda')
params, _ = synth_inputs(64, 32, 2048, 1234, dev)
sigma, gamma, z_far = 0.05, 0.1, 2.0
H = W = 256
p = params[2:3, 1:2].contiguous()
cam = torch.tensor([[1.0, 0.0, 0.0]])
Wd = torch.randn(1, H, W, generator=torch.Generator().manual_seed(5))
kinds = vpn_amd.kinds_tensor([0], dev)
def grads(mask):
    pc = p.cpu().double().requires_grad_(True)
    a, d = O.raster(pc, [0], cam.double(), H, W, sigma, gamma, z_far)
    (d * (Wd * mask).double()).sum().backward()
    pg = p.clone().requires_grad_(True)
    ag, dg = vpn_amd.RasterFunction.apply(pg, kinds, cam.to(dev), H, W, sigma, gamma, z_far)
    (dg * (Wd * mask).to(dev)).sum().backward()
    return pg.grad.cpu().double().flatten(), pc.grad.flatten()
def err(r0, r1, c0, c1):
    m = torch.zeros(1, H, W); m[:, r0:r1, c0:c1] = 1
    g, c = grads(m)
    return float((g - c).abs().max()), g, c
r0, r1, c0, c1 = 0, H, 0, W
while (r1 - r0) > 1 or (c1 - c0) > 1:
    best = None
    rm, cm = (r0 + r1) // 2, (c0 + c1) // 2
    quads = [(a, b, c, d) for (a, b) in ((r0, rm), (rm, r1)) if b > a for (c, d) in ((c0, cm), (cm, c1)) if d > c]
    for qd in quads:
        e, _, _ = err(*qd)
        if best is None or e > best[0]: best = (e, qd)
    print('region', best[1], 'abs err', best[0])
    r0, r1, c0, c1 = best[1]
e, g, c = err(r0, r1, c0, c1)
print('pixel', (r0, c0), 'W', float(Wd[0, r0, c0]), '\n gpu', g.tolist(), '\n ref', c.tolist())
# forward quantities at that pixel from the oracle
pc = p.cpu().double()
a, d = O.raster(pc, [0], cam.double(), H, W, sigma, gamma, z_far)
print('alpha', float(a[0, r0, c0]), 'depth', float(d[0, r0, c0]))
