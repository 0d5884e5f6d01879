import sys, torch, numpy as np
sys.path.insert(0, '.')
import vpn_amd
from oracle import vpn_oracle as O
g = torch.Generator().manual_seed(71)
p1 = torch.rand(2, 63, 3, generator=g) - 0.5
p2 = torch.rand(2, 65, 3, generator=g) - 0.5
d1, i1, d2, i2 = vpn_amd.chamfer_nn(p1.cuda(), p2.cuda())
m1, j1, m2, j2 = O.chamfer_nn(p1, p2)
bad = (d1.cpu() != m1).nonzero()
print('mismatch', bad.shape[0], 'of', m1.numel())
diff = p1[:, :, None, :] - p2[:, None, :, :]
dist = torch.sum(diff * diff, dim=3)
dg = p1.cuda()[:, :, None, :] - p2.cuda()[:, None, :, :]
distg = torch.sum(dg * dg, dim=3)
print('d2 gpu-torch == cpu', torch.equal(distg.cpu(), dist))
sq = dg * dg
distg2 = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
print('d2 gpu explicit == cpu', torch.equal(distg2.cpu(), dist))
print('sqrt gpu-torch == cpu', torch.equal(torch.sqrt(dist.cuda()).cpu(), torch.sqrt(dist)))
for b, i in bad[:5].tolist():
    j = int(j1[b, i])
    x = dist[b, i, j]
    print(b, i, j, 'cpu d2 %r sqrt %r | gpu dist %r | np sqrt %r' % (x.item(), m1[b, i].item(), d1[b, i].item(), float(np.sqrt(np.float32(x.item())))),
          'f64 sqrt', float(np.sqrt(np.float64(x.item()))))
