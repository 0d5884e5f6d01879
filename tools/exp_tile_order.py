"""Does a popcount-sorted launch order of the tiles shorten raster_total_kernel?  The masks of a first call are read
back, the order is built on the host (per image: tiles by visible primitives, heaviest first) and the same call is timed
with and without it (C3 shapes; kernel time from the library's launch profiler)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd
from vpn_amd import _lib
from bench import synth_inputs
dev = torch.device('cuda')
B, K, H, W = int(os.environ.get('B', 64)), 32, 256, 256
params, _ = synth_inputs(B, K, 8, 1234, dev)
p2, _ = synth_inputs(B, K, 8, 4321, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
cam = torch.tensor([[1.0, 0.0, 0.0]], device=dev).expand(B, 3).contiguous()
with torch.no_grad():
    a2, d2 = vpn_amd.RasterFunction.apply(p2, kinds, cam, H, W, 0.05, 0.1, 2.0)
gt_sil, gt_dep = (a2 > 0.5).float().contiguous(), d2.contiguous()
L = _lib.lib()
rec = torch.zeros((L.vpn_raster_records_size(B, K, H, W) // 4,), dtype=torch.float32, device=dev)
lws = torch.zeros((L.vpn_raster_loss_workspace(B, H, W) // 4,), dtype=torch.float32, device=dev)
ws = torch.zeros((L.vpn_raster_bwd_workspace(B, K, H, W) // 4,), dtype=torch.float32, device=dev)
losses = torch.zeros(4, device=dev)


def call(ready, order):
    _lib.call('vpn_raster_total_fwd_fin', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, 0.05, 0.1, 2.0, _lib.ptr(gt_sil),
              _lib.ptr(gt_dep), 0, 1.0, 1.0, _lib.ptr(rec), _lib.ptr(lws), _lib.ptr(ws), ready, None, 0, 0, 0, 0.0, 0.0, 0.0, _lib.ptr(losses),
              None, None, _lib.ptr(order) if order is not None else None, _lib.stream())


call(0, None)
torch.cuda.synchronize()
ref = losses.clone()
gref = ws.clone()
ntile = 256
masks = rec.view(torch.int64)[L.vpn_raster_records_size(B, K, H, W) // 8 - B * ntile:].cpu()
pop = torch.tensor([bin(int(m) & ((1 << 64) - 1)).count('1') for m in masks.tolist()]).reshape(B, ntile)
order = torch.argsort(pop, dim=1, descending=True, stable=True).to(torch.int16).contiguous().to(dev)      # uint16 values < 256


def timed(ready, o, n=30):
    for _ in range(3):
        call(ready, o)
    with _lib.KernelProfile() as kp:
        for _ in range(n):
            call(ready, o)
    return {k: round(v[1] * 1e3, 2) for k, v in kp.summary().items()}


print('position order (masks computed by the tile waves):', timed(0, None))
print('position order, records ready (no record launch)  :', timed(1, None))
print('popcount order, masks read                        :', timed(1, order))
call(1, order)
torch.cuda.synchronize()
print('same losses:', bool(torch.equal(losses, ref)), ' same partials:', bool(torch.equal(ws, gref)))

# where should the tiles without primitives go?  They cost a wave slot for one latency chain and no arithmetic.
def variant(after_pop):
    """descending popcount, the empty tiles moved in front of the tiles with fewer than `after_pop` primitives"""
    out = []
    for b in range(B):
        p = pop[b]
        idx = torch.argsort(p, descending=True, stable=True)
        heavy = idx[p[idx] >= after_pop]
        light = idx[(p[idx] < after_pop) & (p[idx] > 0)]
        empty = idx[p[idx] == 0]
        out.append(torch.cat([heavy, empty, light]))
    return torch.stack(out).to(torch.int16).contiguous().to(dev)


def interleave(every):
    """descending popcount, one empty tile after every `every` non-empty ones (until they run out)"""
    out = []
    for b in range(B):
        p = pop[b]
        idx = torch.argsort(p, descending=True, stable=True)
        ne, em = idx[p[idx] > 0].tolist(), idx[p[idx] == 0].tolist()
        o = []
        while ne or em:
            o += ne[:every]; ne = ne[every:]
            if em:
                o.append(em.pop())
        out.append(torch.tensor(o))
    return torch.stack(out).to(torch.int16).contiguous().to(dev)


def halves(nparts):
    """descending popcount dealt round-robin from `nparts` equal parts of the sorted list: H M L H M L ..."""
    out = []
    for b in range(B):
        idx = torch.argsort(pop[b], descending=True, stable=True).tolist()
        parts = [idx[i * ntile // nparts:(i + 1) * ntile // nparts] for i in range(nparts)]
        o = []
        for i in range(max(len(p) for p in parts)):
            for p in parts:
                if i < len(p):
                    o.append(p[i])
        out.append(torch.tensor(o))
    return torch.stack(out).to(torch.int16).contiguous().to(dev)


def head_then_mix(frac):
    """the heaviest `frac` of the tiles first (descending), the rest heavy / light alternating"""
    out = []
    for b in range(B):
        idx = torch.argsort(pop[b], descending=True, stable=True).tolist()
        h = int(ntile * frac)
        head, rest = idx[:h], idx[h:]
        a, c = rest[:len(rest) // 2], rest[len(rest) // 2:][::-1]
        o = list(head)
        for i in range(max(len(a), len(c))):
            if i < len(a):
                o.append(a[i])
            if i < len(c):
                o.append(c[i])
        out.append(torch.tensor(o))
    return torch.stack(out).to(torch.int16).contiguous().to(dev)


g = torch.Generator().manual_seed(1)
shuffled = torch.stack([torch.randperm(ntile, generator=g) for _ in range(B)]).to(torch.int16).contiguous().to(dev)
variants = (('desc (rider)', order), ('dealt from 2 halves', halves(2)), ('dealt from 3 parts', halves(3)), ('dealt from 5 parts', halves(5)),
            ('heaviest 10% then mix', head_then_mix(0.1)), ('heaviest 30% then mix', head_then_mix(0.3)), ('random', shuffled))
res = {name: [] for name, _ in variants}
for rnd in range(6):                       # round-robin: the box's clock drifts by several % over a run
    for name, o in variants:
        res[name].append(timed(1, o, 40)['raster_total_kernel'])
for name, _ in variants:
    v = sorted(res[name][1:])
    print('%-24s median %.2f  min %.2f  all %s' % (name, v[len(v) // 2], v[0], res[name]))
