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
# the tile entries come from the rider of the scan's launch (tile, popcount, mask, quadrant masks); a launch order is a
# permutation of the by-tile half of its buffer
n, M = 256, 2048
N = K * n
gt_points = torch.rand(B, M, 3, device=dev) - 0.5
cws = torch.zeros((L.vpn_chamfer_workspace(B, N, M) // 4,), dtype=torch.float32, device=dev)
pts = torch.empty(B, N, 3, device=dev)
d1, d2 = torch.empty(B, N, device=dev), torch.empty(B, M, device=dev)
i1, i2 = torch.empty(B, N, dtype=torch.int32, device=dev), torch.empty(B, M, dtype=torch.int32, device=dev)
ent = torch.zeros((L.vpn_raster_order_size(B, H, W) // 8,), dtype=torch.int64, device=dev)
_lib.call('vpn_hotpath_sample_fwd', _lib.ptr(params), _lib.ptr(kinds), None, 1234, None, 0, B, K, n, _lib.ptr(pts), _lib.ptr(cam), H, W, 0.05,
          _lib.ptr(rec), _lib.ptr(lws), _lib.ptr(gt_points), M, _lib.ptr(cws), cws.numel() * 4, _lib.stream())
_lib.call('vpn_hotpath_chamfer_fwd', _lib.ptr(pts), _lib.ptr(gt_points), B, N, M, _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2), _lib.ptr(i2),
          _lib.ptr(cws), cws.numel() * 4, 7, _lib.ptr(rec), K, H, W, _lib.ptr(ent), _lib.stream())
torch.cuda.synchronize()
by_tile = ent.view(2, B, ntile, 6)[1].clone()


def entries_of(perm):
    """perm [B][ntile] (tile of every rank) -> an entries buffer the raster takes"""
    e = torch.zeros_like(ent).view(2, B, ntile, 6)
    e[0] = torch.gather(by_tile, 1, perm.to(dev).long()[:, :, None].expand(B, ntile, 6))
    return e.reshape(-1).contiguous()


order = ent.clone()                                    # the rider's own order (descending popcount)


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
    return entries_of(torch.stack(out))


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
    return entries_of(torch.stack(out))


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
    return entries_of(torch.stack(out))


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
    return entries_of(torch.stack(out))


g = torch.Generator().manual_seed(1)
shuffled = entries_of(torch.stack([torch.randperm(ntile, generator=g) for _ in range(B)]))
variants = (('desc (rider)', order), ('dealt from 2 halves', halves(2)), ('dealt from 3 parts', halves(3)), ('dealt from 5 parts', halves(5)),
            ('heaviest 10% then mix', head_then_mix(0.1)), ('heaviest 30% then mix', head_then_mix(0.3)), ('random', shuffled))
res = {name: [] for name, _ in variants}
for rnd in range(6):                       # round-robin: the box's clock drifts by several % over a run
    for name, o in variants:
        res[name].append(timed(1, o, 40)['raster_total_kernel'])
for name, _ in variants:
    v = sorted(res[name][1:])
    print('%-24s median %.2f  min %.2f  all %s' % (name, v[len(v) // 2], v[0], res[name]))
