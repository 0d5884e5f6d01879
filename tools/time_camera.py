"""Fused camera transform vs the chain of rotate launches it replaces (C3 cloud)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd
from vpn_amd.modules import transform as T
B, N = 64, 8192
g = torch.Generator().manual_seed(0)
pts = (torch.rand(B, N, 3, generator=g) - 0.5).cuda().requires_grad_(True)
d = (torch.rand(B, generator=g) + 0.8).cuda(); e = (torch.rand(B, generator=g) * 60).cuda()
a = (torch.rand(B, generator=g) * 360).cuda(); ang = (torch.rand(B, generator=g) * 360).cuda()

def chain(points):          # what modules/transform.py did before the fused kernel (= the reference's structure)
    e1, a1 = e.view(-1, 1) / 360, a.view(-1, 1) / 360
    points = T.rotate_points_forward_x_axis(points, -ang)
    y = torch.tensor([[0.0, 1.0, 0.0]], device=points.device).repeat(B, 1)
    y = T.rotate_points(y.unsqueeze(1), T._axis_q([0.0, 0.0, -1.0], e1)).squeeze(1)
    points = T.rotate_points(points, torch.cat([y, -a1], 1))
    points = T.rotate_points(points, T._axis_q([0.0, 0.0, -1.0], -e1))
    return points * d.view(-1, 1, 1)

def bench(f, tag):
    for _ in range(5):
        out = f(pts); out.sum().backward()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50):
        out = f(pts); pts.grad = None; out.sum().backward()
    torch.cuda.synchronize()
    print('%s fwd+bwd: %.1f us' % (tag, (time.perf_counter() - t) / 50 * 1e6), flush=True)
    return out
o1 = bench(chain, 'chain of 4 rotate launches + scale')
o2 = bench(lambda p: T.view_to_obj_points(p, d, e, a, ang), 'fused camera transform')
print('max diff', (o1 - o2).abs().max().item())
