"""Timeline of one chamfer_nn_mfma_kernel<2> launch of the C3 step (library built with -DCM_EXP_TRACE): per scan workgroup
start / end of the tile loop / end of the per-query finish / end, and the resident workgroups over time."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd
from vpn_amd import _lib
from bench import synth_inputs
dev = torch.device('cuda')
B, K, n, M, H, W = int(os.environ.get('B', 64)), 32, 256, 2048, 256, 256
params, gt_points = synth_inputs(B, K, M, 1234, dev)
p2, _ = synth_inputs(B, K, 8, 4321, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
cam = torch.tensor([[1.0, 0.0, 0.0]], device=dev).expand(B, 3).contiguous()
with torch.no_grad():
    a2, d2 = vpn_amd.RasterFunction.apply(p2, kinds, cam, H, W, 0.05, 0.1, 2.0)
gs, gd = (a2 > 0.5).float().contiguous(), d2.contiguous()
params.requires_grad_(True)
seed = torch.full((1,), 1234, dtype=torch.int64, device=dev)
one = torch.ones((), device=dev)
for i in range(60):                                     # warm clocks
    params.grad = None
    o = vpn_amd.HotPathLossFunction.apply(params, kinds, cam, gt_points, gs, gd, n, seed, 0, H, W, 0.05, 0.1, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, False, True)
    o[2].backward(one)
torch.cuda.synchronize()
nwg = B * (M // 256 + K * n // 256)
buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
L = _lib.lib()
L.vpn_debug_scan_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.vpn_debug_scan_trace(ctypes.c_void_p(buf.data_ptr()), nwg) == 0
torch.cuda.synchronize()
t = buf.cpu().reshape(nwg, 8)
t0 = int(t[:, 0].min())
us = lambda c: (t[:, c] - t0).float() / 100.0
start, loop, epi, end, job, cnt = us(0), us(1), us(2), us(3), t[:, 4], t[:, 5]
print('launch: last end %.1f us; %d workgroups (job 0 = the long direction, %d of them)' % (float(end.max()), nwg, int((job == 0).sum())))
for j in (0, 1):
    m = job == j
    print('job %d: %4d workgroups | loop %.1f us (min %.1f max %.1f) | finish %.2f us | fix-up + sums %.2f us (undecided per workgroup %.2f) | starts %.1f..%.1f, last end %.1f' % (
        j, int(m.sum()), float((loop - start)[m].mean()), float((loop - start)[m].min()), float((loop - start)[m].max()), float((epi - loop)[m].mean()),
        float((end - epi)[m].mean()), float(cnt[m].float().mean()), float(start[m].min()), float(start[m].max()), float(end[m].max())))
for x in range(0, int(end.max()) + 4, 4):
    live = (start <= x) & (end > x)
    print('t = %3d us: %4d workgroups resident (%3d long), %4d of them in the finish / fix-up' % (x, int(live.sum()), int((live & (job == 0)).sum()), int((live & (loop <= x)).sum())))
