"""Is the raster backward exactly linear under W -> 2 W?  Calls vpn_raster_bwd on the C3 raster case of
test_raster_full_size_properties with its own workspaces and compares them entry by entry (finding: one tile partial in
6.3 M differs by one ulp, with either reduction of the tile sums; the workspace fill value does not matter).  With a library
built with VPN_RASTER_FLAGS=-DR_DEBUG_LIN it then re-runs both backward passes with the differing (image, primitive, tile)
selected for a dump of every lane's sums and per-pixel intermediates, and prints where 2 x value(W) != value(2 W) first
appears."""
import sys, torch
sys.path.insert(0, '/root/repo')
import vpn_amd as vpn
from vpn_amd import _lib
DEV = 'cuda'
g = lambda t: t.to(DEV)
def rand_params(gen, B, K):
    v = (torch.rand(B, K, 3, generator=gen) + 0.1) / torch.tensor([8.0, 10.0, 10.0])
    q = torch.rand(B, K, 4, generator=gen)
    t = 0.35 * (torch.rand(B, K, 3, generator=gen) * 2 - 1)
    return torch.cat([v, q, t], 2)
gen = torch.Generator().manual_seed(1234)
B, K, H, W = 64, 32, 256, 256
params = g(rand_params(gen, B, K))
kinds = vpn.kinds_tensor([0] * K, torch.device(DEV))
cam = g(torch.tensor([[1.0, 0.0, 0.0]]).expand(B, 3).contiguous())
L = _lib.lib()
alpha, depth = torch.empty(B, H, W, device=DEV), torch.empty(B, H, W, device=DEV)
aux = torch.empty(B, 3, H, W, device=DEV)
rec = torch.zeros(L.vpn_raster_records_size(B, K, H, W) // 4, device=DEV)
_lib.call('vpn_raster_fwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, 0.05, 0.1, 2.0, _lib.ptr(alpha), _lib.ptr(depth), _lib.ptr(aux), _lib.ptr(rec), _lib.stream())
Wa, Wd = g(torch.randn(B, H, W, generator=gen)), g(torch.randn(B, H, W, generator=gen))
nws = L.vpn_raster_bwd_workspace(B, K, H, W) // 4
def bwd(sc, fill):
    ws = torch.full((nws,), fill, device=DEV)
    gp = torch.empty_like(params)
    ga, gd = (sc * Wa).contiguous(), (sc * Wd).contiguous()
    _lib.call('vpn_raster_bwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, 0.05, 0.1, 2.0, _lib.ptr(aux), _lib.ptr(rec), _lib.ptr(ga), _lib.ptr(gd),
              _lib.ptr(ws), _lib.ptr(gp), _lib.stream())
    torch.cuda.synchronize()
    return ws, gp
w1, g1 = bwd(1.0, 0.0)
w1n, g1n = bwd(1.0, float('nan'))
w2, g2 = bwd(2.0, 0.0)
print('g with zero-filled vs NaN-filled workspace equal:', bool(torch.equal(g1, g1n)), ' NaNs in g:', int(torch.isnan(g1n).sum()))
d = (w2 - 2 * w1).abs()
print('workspace floats', nws, 'entries where ws(2W) != 2 ws(W):', int((d > 0).sum()), 'max', float(d.max()))
dg = (g2 - 2 * g1).abs()
print('grad entries differing:', int((dg > 0).sum()), [tuple(i) for i in (dg > 0).nonzero()[:4].tolist()])
idx = (d > 0).nonzero().flatten()[:10].tolist()
ntile = 256
for i in idx:
    bk, rem = divmod(i, ntile * 12); tile, c = divmod(rem, 12)
    print('   ws index', i, '(b,k)=', divmod(bk, K), 'tile', tile, 'component', c, float(w1[i]), float(w2[i]))

import ctypes
if idx and hasattr(L, 'vpn_debug_raster_lin'):
    L.vpn_debug_raster_lin.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    i = idx[0]
    bk, rem = divmod(i, ntile * 12); tile, c = divmod(rem, 12); b_, k_ = divmod(bk, K)
    dumps = []
    for sc in (1.0, 2.0):
        buf = torch.zeros(64 * 16 + 64 * 4 * 16, device=DEV)
        assert L.vpn_debug_raster_lin(b_, k_, tile, ctypes.c_void_p(buf.data_ptr())) == 0
        bwd(sc, 0.0)
        dumps.append(buf.cpu().clone())
    L.vpn_debug_raster_lin(-1, -1, -1, None)
    d1, d2 = dumps
    lanes1, lanes2 = d1[:64 * 16].reshape(64, 16), d2[:64 * 16].reshape(64, 16)
    bad = (lanes2 != 2 * lanes1).nonzero()
    print('lane sums (64 x 12) where sum(2W) != 2 sum(W):', bad.tolist()[:8])
    names = ['gw', 'gz', 'ga', 'gx', 'gm2', 'go0', 'go1', 'go2', 'gd0', 'gd1', 'gd2', 'wgt', 'E', 'a', 'gAtot', 'gZbar']
    p1, p2 = d1[64 * 16:].reshape(64, 4, 16), d2[64 * 16:].reshape(64, 4, 16)
    lin = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 15]                    # the quantities that are linear in W
    badp = (p2[:, :, lin] != 2 * p1[:, :, lin]).nonzero()
    print('per-pixel intermediates where value(2W) != 2 value(W):', len(badp))
    tiny = 1.1754944e-38
    for lane_, s_, j in badp.tolist()[:12]:
        n = names[lin[j]]
        print('   lane %2d slot %d %-6s W: %.9e  2W: %.9e   (2^-126 = %.3e; the row of W: %s)' % (
            lane_, s_, n, float(p1[lane_, s_, lin[j]]), float(p2[lane_, s_, lin[j]]), tiny,
            ' '.join('%s=%.3e' % (names[q], float(p1[lane_, s_, q])) for q in (0, 1, 2, 3, 4, 11, 12, 13, 14, 15))))
    # the lanes whose SUMS differ: their four pixels in full, W next to 2 W / 2 (hex: every bit)
    hx = lambda v: float(v).hex()
    for lane_, comp in bad.tolist()[:4]:
        print('   lane %2d component %2d: sum(W) %s   sum(2W)/2 %s' % (lane_, comp, hx(lanes1[lane_, comp]), hx(lanes2[lane_, comp] / 2)))
        for s_ in range(4):
            print('      slot %d: a=%.3e wgt=%.3e  ' % (s_, float(p1[lane_, s_, 13]), float(p1[lane_, s_, 11])) +
                  '  '.join('%s %s | %s' % (names[q], hx(p1[lane_, s_, q]), hx(p2[lane_, s_, q] / 2)) for q in (1, 4, 5, 6, 7, 8, 9, 10)))
    if not len(badp) and len(bad):
        print('every per-pixel intermediate doubles exactly: the difference arises in the accumulation of the lane sums (v += ...)')
        for lane_, comp in bad.tolist()[:6]:
            print('   lane %2d component %2d: W %.9e  2W %.9e ; its four pixels (go/gd terms, W):' % (lane_, comp, float(lanes1[lane_, comp]), float(lanes2[lane_, comp])),
                  [[float(x) for x in p1[lane_, s_, 5:11]] for s_ in range(4)])
else:
    print('no differing partial, or the library was not built with -DR_DEBUG_LIN')
