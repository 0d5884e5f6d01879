"""bench.py's parity leg by loss component: HotPathLossFunction vs the oracle with weights (cd, sil, depth) switched on
one at a time, GT images rendered by the oracle or by the HIP raster."""
import sys, torch
sys.path.insert(0, '.')
import vpn_amd
from oracle import vpn_oracle as O
from bench import synth_inputs
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda')
K, n, M, H, W = 32, 256, 2048, 256, 256
sigma, gamma, z_far = vpn_amd.config.RASTER_SIGMA, vpn_amd.config.RASTER_GAMMA, vpn_amd.config.RASTER_Z_FAR
params, gt_points = synth_inputs(64, K, M, 1234, 'cpu')
p2, _ = synth_inputs(64, K, M, 4321, 'cpu')
params, gt_points, p2 = params[:S], gt_points[:S], p2[:S]
kl = [0] * K
kinds = vpn_amd.kinds_tensor(kl, dev)
cam = torch.tensor([[1.0, 0.0, 0.0]]).expand(S, 3).contiguous()
torch.set_num_threads(16)
u = O.philox_uniforms(1234, 0, S, K, n)
with torch.no_grad():
    r = [O.raster(p2[b:b + 1], kl, cam[b:b + 1], H, W, sigma, gamma, z_far) for b in range(S)]
    gt_o = (torch.cat([x[0] for x in r]) > 0.5).float(), torch.cat([x[1] for x in r])
    a2, d2 = vpn_amd.RasterFunction.apply(p2.to(dev), kinds, cam.to(dev), H, W, sigma, gamma, z_far)
    gt_h = (a2 > 0.5).float().cpu(), d2.cpu()
print('GT sil pixels that differ oracle vs HIP: %d ; max |depth diff| %.2e' % (int((gt_o[0] != gt_h[0]).sum()), float((gt_o[1] - gt_h[1]).abs().max())))
def cpu(w, gt, dt=torch.float32):
    p = params.detach().clone().to(dt).requires_grad_(True)
    tot = 0.0
    for b in range(S):
        pb = p[b:b + 1]
        loss = 0.0
        if w[0]:
            pts = O.sample_primitives(pb, kl, u[b:b + 1].to(dt))
            loss = loss + w[0] * O.chamfer_loss(pts, gt_points[b:b + 1].to(dt), each_batch=True).sum() / S
        if w[1] or w[2]:
            a, d = O.raster(pb, kl, cam[b:b + 1].to(dt), H, W, sigma, gamma, z_far)
            loss = loss + w[1] * (a - gt[0][b:b + 1].to(dt)).abs().sum() / (S * H * W) + w[2] * (d - gt[1][b:b + 1].to(dt)).abs().sum() / (S * H * W)
        loss.backward()
        tot += float(loss.detach())
    return tot, p.grad
def gpu(w, gt):
    pg = params.to(dev).requires_grad_(True)
    out = vpn_amd.HotPathLossFunction.apply(pg, kinds, cam.to(dev), gt_points.to(dev), gt[0].to(dev), gt[1].to(dev), n, 1234, 0, H, W,
                                            sigma, gamma, z_far, *w)
    out[2].backward()
    return float(out[2]), pg.grad.cpu()
def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())
for gname, gt in (('oracle GT', gt_o), ('HIP GT', gt_h)):
    for w in ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), (1.0, 1.0, 1.0)):
        lg, gg = gpu(w, gt)
        lc, gc = cpu(w, gt)
        l64, g64 = cpu(w, gt, torch.float64)
        e = (gg.double() - g64).abs(); i = int(e.flatten().argmax())
        print('%-9s w=%s loss rel %.1e | grad gpu-vs-cpu32 %.2e gpu-vs-cpu64 %.2e cpu32-vs-cpu64 %.2e | worst (b,k,c)=(%d,%d,%d)'
              % (gname, w, abs(lg - lc) / abs(lc), rel(gg, gc), rel(gg, g64), rel(gc, g64), i // (K * 10), i // 10 % K, i % 10))
