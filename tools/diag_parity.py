"""Per-component gradient parity on the first S images of the bench workload:
HIP fp32 vs oracle fp32 vs oracle fp64."""
import sys, torch
sys.path.insert(0, '.')
import vpn_amd
from oracle import vpn_oracle as O
from bench import synth_inputs
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
B, K, n, M, H, W = 64, 32, 256, 2048, 256, 256
sigma, gamma, z_far = vpn_amd.config.RASTER_SIGMA, vpn_amd.config.RASTER_GAMMA, vpn_amd.config.RASTER_Z_FAR
params, gt = synth_inputs(B, K, M, 1234, dev)
p2, _ = synth_inputs(B, K, M, 4321, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
cam = torch.tensor([[1.0, 0.0, 0.0]], device=dev).expand(B, 3).contiguous()
with torch.no_grad():
    a2, d2 = vpn_amd.RasterFunction.apply(p2, kinds, cam, H, W, sigma, gamma, z_far)
gt_sil, gt_depth = (a2 > 0.5).float(), d2.clone()
torch.set_num_threads(16)
Wd = torch.randn(B, H, W, generator=torch.Generator().manual_seed(5)).to(dev)
u = O.philox_uniforms(1234, 0, S, K, n)
kl = [0] * K
def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())
def gpu(which):
    p = params[:S].detach().clone().requires_grad_(True)
    if which == 'cd':
        pts = vpn_amd.Sampling.sample_primitives(p, kinds, n, seed=1234)
        loss = vpn_amd.ChamferDistanceLoss()(pts, gt[:S])
    else:
        a, d = vpn_amd.RasterFunction.apply(p, kinds, cam[:S].contiguous(), H, W, sigma, gamma, z_far)
        loss = (a - gt_sil[:S]).abs().mean() if which == 'sil' else ((d - gt_depth[:S]).abs().mean() if which == 'depth' else ((d * Wd[:S]).sum() if which == 'depthw' else (a * Wd[:S]).sum()))
    loss.backward()
    return p.grad.cpu()
def cpu(which, dt):
    p = params[:S].detach().cpu().to(dt).requires_grad_(True)
    for b in range(S):
        pb = p[b:b + 1]
        if which == 'cd':
            pts = O.sample_primitives(pb, kl, u[b:b + 1].to(dt))
            loss = O.chamfer_loss(pts, gt[b:b + 1].cpu().to(dt), each_batch=True).sum() / S
        else:
            a, d = O.raster(pb, kl, cam[b:b + 1].cpu().to(dt), H, W, sigma, gamma, z_far)
            if which in ('depthw', 'alphaw'):
                loss = ((d if which == 'depthw' else a) * Wd[b:b + 1].cpu().to(dt)).sum()
            else:
                loss = ((a - gt_sil[b:b + 1].cpu().to(dt)).abs().sum() if which == 'sil' else (d - gt_depth[b:b + 1].cpu().to(dt)).abs().sum()) / (S * H * W)
        loss.backward()
    return p.grad
for which in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("cd", "sil", "depth")):
    g = gpu(which); c32 = cpu(which, torch.float32); c64 = cpu(which, torch.float64)
    print('%-6s |grad|max %.3e  gpu-vs-cpu32 %.2e  gpu-vs-cpu64 %.2e  cpu32-vs-cpu64 %.2e' % (which, float(c64.abs().max()), rel(g, c32), rel(g, c64), rel(c32, c64)))
    e = (g.double() - c64).abs()
    i = e.flatten().argmax(); b, k, j = int(i // (K * 10)), int(i // 10 % K), int(i % 10)
    print('       worst (b,k,comp)=(%d,%d,%d) gpu %.6e cpu32 %.6e cpu64 %.6e  params %s' % (b, k, j, g[b, k, j], c32[b, k, j], c64[b, k, j], params[b, k].cpu().tolist()))
