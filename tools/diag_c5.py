import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import vpn_amd
from oracle import vpn_oracle as O
from conftest import decidable_depth_gt
from test_gpu_parity import rand_params
gen = torch.Generator().manual_seed(56)
B, K, n, M, H, W = 2, 64, 128, 2048, 256, 256
params = rand_params(gen, B, K)
kinds = [0] * K
gt_pts = torch.rand(B, M, 3, generator=gen) - 0.5
gt_sil = (torch.rand(B, 1, H, W, generator=gen) > 0.5).float()
cam = torch.tensor([[1.0, 0.0, 0.0]]).expand(B, 3).contiguous()
gt_dep = decidable_depth_gt(O, params, kinds, cam, 2.0 - torch.rand(B, H, W, generator=gen), H, W, chunk=1)
dev = torch.device('cuda')
kt = vpn_amd.kinds_tensor(kinds, dev)
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
for w in ((0., 1., 0.), (0., 0., 1.), (1., 0., 0.), (1., 1., 1.)):
    res = {}
    for dt in (torch.float32, torch.float64):
        pc = params.to(dt).clone().requires_grad_(True)
        u = O.philox_uniforms(99, 0, B, K, n).to(dt)
        for b in range(B):
            loss = 0
            if w[0]:
                loss = loss + w[0] * O.chamfer_loss(O.sample_primitives(pc[b:b+1], kinds, u[b:b+1]), gt_pts[b:b+1].to(dt)) / B
            if w[1] or w[2]:
                a, d = O.raster(pc[b:b+1], kinds, cam[b:b+1].to(dt), H, W, 0.05, 0.1, 2.0)
                loss = loss + w[1] * (a[:, None] - gt_sil[b:b+1].to(dt)).abs().sum() / (B*H*W) + w[2] * (d - gt_dep[b:b+1].to(dt)).abs().sum() / (B*H*W)
            loss.backward()
        res[dt] = pc.grad
    pg = params.to(dev).requires_grad_(True)
    out = vpn_amd.HotPathLossFunction.apply(pg, kt, cam.to(dev), gt_pts.to(dev), gt_sil.to(dev), gt_dep.to(dev), n, 99, 0, H, W, 0.05, 0.1, 2.0, *w)
    out[2].backward()
    g = pg.grad.cpu()
    e = (g.double() - res[torch.float64]).abs(); i = int(e.flatten().argmax())
    print(w, 'gpu-vs-cpu32 %.2e gpu-vs-cpu64 %.2e cpu32-vs-cpu64 %.2e worst (b,k,c)=(%d,%d,%d) gpu %.6e c32 %.6e c64 %.6e' % (rel(g, res[torch.float32]), rel(g, res[torch.float64]), rel(res[torch.float32], res[torch.float64]), i // (K*10), i // 10 % K, i % 10, g.flatten()[i], res[torch.float32].flatten()[i], res[torch.float64].flatten()[i]))
