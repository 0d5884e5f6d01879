"""Is the L1 depth loss's gradient decidable in fp32?  sign(D - gt_depth) flips where |D - gt| is below rounding
noise.  Counts such pixels on the bench's parity sample with an ORACLE-rendered GT, and compares the depth-loss
gradient HIP / oracle fp32 / oracle fp64 with the raw GT and with a GT moved by 1e-2 at the near-tie pixels."""
import sys, torch
sys.path.insert(0, '.')
import vpn_amd
from oracle import vpn_oracle as O
from bench import synth_inputs
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda')
K, M, H, W = 32, 2048, 256, 256
sigma, gamma, z_far = vpn_amd.config.RASTER_SIGMA, vpn_amd.config.RASTER_GAMMA, vpn_amd.config.RASTER_Z_FAR
params, _ = synth_inputs(64, K, M, 1234, 'cpu')
p2, _ = synth_inputs(64, K, M, 4321, 'cpu')
params, p2 = params[:S], p2[:S]
kl = [0] * K
kinds = vpn_amd.kinds_tensor(kl, dev)
cam = torch.tensor([[1.0, 0.0, 0.0]]).expand(S, 3).contiguous()
torch.set_num_threads(16)
with torch.no_grad():
    gt = torch.cat([O.raster(p2[b:b + 1], kl, cam[b:b + 1], H, W, sigma, gamma, z_far)[1] for b in range(S)])
    dp = torch.cat([O.raster(params[b:b + 1], kl, cam[b:b + 1], H, W, sigma, gamma, z_far)[1] for b in range(S)])
diff = (dp - gt).abs()
for thr in (1e-6, 1e-5, 1e-4):
    print('pixels with 0 < |D - gt| < %g: %d   (exact ties %d of %d)' % (thr, int(((diff > 0) & (diff < thr)).sum()), int((diff == 0).sum()), diff.numel()))
near = (diff > 0) & (diff < 1e-4)
gt_fixed = torch.where(near, gt + torch.where(dp >= gt, -1e-2, 1e-2), gt)
def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())
def gpu(g):
    p = params.to(dev).requires_grad_(True)
    a, d = vpn_amd.RasterFunction.apply(p, kinds, cam.to(dev), H, W, sigma, gamma, z_far)
    (d - g.to(dev)).abs().mean().backward()
    return p.grad.cpu()
def cpu(g, dt):
    p = params.detach().clone().to(dt).requires_grad_(True)
    for b in range(S):
        a, d = O.raster(p[b:b + 1], kl, cam[b:b + 1].to(dt), H, W, sigma, gamma, z_far)
        ((d - g[b:b + 1].to(dt)).abs().sum() / (S * H * W)).backward()
    return p.grad
for name, g in (('raw GT', gt), ('near-ties moved', gt_fixed)):
    a, c32, c64 = gpu(g), cpu(g, torch.float32), cpu(g, torch.float64)
    print('%-16s gpu-vs-cpu32 %.2e  gpu-vs-cpu64 %.2e  cpu32-vs-cpu64 %.2e' % (name, rel(a, c32), rel(a, c64), rel(c32, c64)))
