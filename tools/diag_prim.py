import sys, torch
sys.path.insert(0, '.')
import vpn_amd
from oracle import vpn_oracle as O
from bench import synth_inputs
dev = torch.device('cuda')
params, _ = synth_inputs(64, 32, 2048, 1234, dev)
sigma, gamma, z_far = 0.05, 0.1, 2.0
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
def run(p, H, W, tag):
    K = p.shape[1]
    kinds = [0] * K
    cam = torch.tensor([[1.0, 0.0, 0.0]])
    Wd = torch.randn(1, H, W, generator=torch.Generator().manual_seed(5))
    pc = p.cpu().double().requires_grad_(True)
    a, d = O.raster(pc, kinds, cam.double(), H, W, sigma, gamma, z_far)
    (d * Wd.double()).sum().backward()
    pg = p.clone().requires_grad_(True)
    ag, dg = vpn_amd.RasterFunction.apply(pg, vpn_amd.kinds_tensor(kinds, dev), cam.to(dev), H, W, sigma, gamma, z_far)
    (dg * Wd.to(dev)).sum().backward()
    e = (pg.grad.cpu().double() - pc.grad).abs()
    i = int(e.flatten().argmax())
    print('%-28s fwd depth rel %.2e alpha rel %.2e | grad rel %.2e  worst k=%d comp=%d gpu %.5e ref %.5e' % (
        tag, rel(dg.cpu(), d), rel(ag.cpu(), a), rel(pg.grad.cpu(), pc.grad), i // 10, i % 10, pg.grad.cpu().flatten()[i], pc.grad.flatten()[i]))
    return dg.detach().cpu(), d.detach()
p = params[2:3]
for H in (64, 128, 256):
    run(p[:, 1:2].contiguous(), H, H, 'prim (2,1) alone %d' % H)
run(p.contiguous(), 256, 256, 'image 2 all prims 256')
run(p[:, :8].contiguous(), 256, 256, 'image 2 prims 0..7 256')
run(p[:, :2].contiguous(), 256, 256, 'image 2 prims 0..1 256')
