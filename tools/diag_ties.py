import sys, torch
sys.path.insert(0, '.')
import vpn_amd
from bench import synth_inputs
dev = torch.device('cuda')
B, K, n, M = 64, 32, 256, 2048
params, gt = synth_inputs(B, K, M, 1234, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
pts = vpn_amd.Sampling.sample_primitives(params, kinds, n, seed=1234)
def ties(q, t):
    tot = 0; dup = 0
    for b in range(q.shape[0]):
        d = q[b][:, None, :] - t[b][None, :, :]
        dd = d * d
        d2 = (dd[..., 0] + dd[..., 1]) + dd[..., 2]
        m, j0 = d2.min(1)
        j0 = (d2 == m[:, None]).float().argmax(1)          # first index attaining the min
        cm = torch.cummin(d2, 1).values
        prev = torch.where(j0 > 0, cm.gather(1, (j0 - 1).clamp_min(0)[:, None])[:, 0], torch.full_like(m, float('inf')))
        tie = torch.sqrt(prev) == torch.sqrt(m)
        tot += int(tie.sum())
        if tie.any():
            i = int(tie.nonzero()[0])
            print('b', b, 'query', i, 'm', m[i].item(), 'prev', prev[i].item(), 'j0', int(j0[i]), 'ulps', (prev[i].view(torch.int32) - m[i].view(torch.int32)).item())
    return tot
print('p2->p1 ties (queries=gt):', ties(gt, pts))
print('p1->p2 ties (queries=pred):', ties(pts, gt))
