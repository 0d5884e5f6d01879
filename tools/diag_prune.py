import sys, ctypes, torch
sys.path.insert(0, '.')
import vpn_amd
from vpn_amd import _lib
from bench import synth_inputs
dev = torch.device('cuda')
B, K, n, M = 64, 32, 256, 2048
params, gt = synth_inputs(B, K, M, 1234, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
pts = vpn_amd.Sampling.sample_primitives(params, kinds, n, seed=1234)
L = _lib.lib()
buf = (ctypes.c_ulonglong * 8)()
def rd():
    torch.cuda.synchronize(); L.vpn_debug_read(buf); return list(buf)
def run(q, t, tag):
    a = rd()
    Bq, Nq, Nt = q.shape[0], q.shape[1], t.shape[1]
    d = torch.empty(Bq, Nq, device=dev); i = torch.empty(Bq, Nq, dtype=torch.int32, device=dev)
    d2 = torch.empty(Bq, Nt, device=dev); i2 = torch.empty(Bq, Nt, dtype=torch.int32, device=dev)
    ws = torch.empty(L.vpn_chamfer_workspace(Bq, Nq, Nt) // 4, device=dev)
    _lib.call('vpn_chamfer_fwd_ws', _lib.ptr(q), _lib.ptr(t), Bq, Nq, Nt, _lib.ptr(d), _lib.ptr(i), _lib.ptr(d2), _lib.ptr(i2), _lib.ptr(ws), ws.numel() * 4, 2, _lib.stream())
    b = rd()
    waves1, waves2 = Bq * ((Nq + 63) // 64), Bq * ((Nt + 63) // 64)
    c1, c2 = (Nt + 63) // 64, (Nq + 63) // 64
    print(tag, 'chunks processed (both directions) %d of max %d ; tie rescans %d' % (b[5] - a[5], waves1 * c1 + waves2 * c2, b[4] - a[4]))
run(pts, gt, 'synthetic uniform GT   :')
# a "realistic" case: GT points ON the surface of a second primitive set close to the prediction
p2 = params.clone(); p2[..., 7:] += 0.02 * torch.randn_like(p2[..., 7:])
gt2 = vpn_amd.Sampling.sample_primitives(p2, kinds, 64, seed=99)       # 2048 points on nearby primitives
run(pts, gt2, 'GT on nearby surfaces  :')
