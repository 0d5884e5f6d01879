"""Timeline of one raster_total_kernel launch of the C3 step (library built with -DR_EXP_TRACE): per tile wave start / end
(100 MHz wall clock) and visible primitives.  Prints how long tiles of each weight live, when the waves of each weight
class end, and the number of resident waves over time."""
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
for i in range(30):                                     # warm clocks
    params.grad = None
    o = vpn_amd.HotPathLossFunction.apply(params, kinds, cam, gt_points, gs, gd, n, seed, 0, H, W, 0.05, 0.1, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, False, True)
    o[2].backward(one)
torch.cuda.synchronize()
nw = B * 256
buf = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
L = _lib.lib()
L.vpn_debug_raster_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.vpn_debug_raster_trace(ctypes.c_void_p(buf.data_ptr()), nw) == 0
torch.cuda.synchronize()
t = buf.cpu().reshape(nw, 8)
t0 = int(t[:, 0].min())
start, end, pop, hw = (t[:, 0] - t0).float() / 100.0, (t[:, 1] - t0).float() / 100.0, t[:, 2], t[:, 3]
print('launch: first start 0, last end %.1f us; waves %d' % (float(end.max()), nw))
for c in range(0, 11):
    m = pop == c
    if int(m.sum()):
        d = (end - start)[m]
        print('popcount %2d: %5d tiles, life mean %5.1f us (min %5.1f, max %5.1f), starts %5.1f..%5.1f, last end %5.1f' % (
            c, int(m.sum()), float(d.mean()), float(d.min()), float(d.max()), float(start[m].min()), float(start[m].max()), float(end[m].max())))
ph = [(t[:, 4] - t[:, 0]).float() / 100.0, (t[:, 5] - t[:, 4]).float() / 100.0, (t[:, 6] - t[:, 5]).float() / 100.0, (t[:, 1] - t[:, 6]).float() / 100.0]
print('phases (mean us): prologue + forward | losses + publish | backward | arrival wait + end')
for c in range(0, 11):
    m = pop == c
    if int(m.sum()):
        print('popcount %2d: ' % c + ' | '.join('%5.2f' % float(x[m].mean()) for x in ph))
late = start > 30.0
for c in (0, 1, 2):
    m = (pop == c) & late
    if int(m.sum()):
        print('popcount %2d started after 30 us: ' % c + ' | '.join('%5.2f' % float(x[m].mean()) for x in ph))
for x in range(0, int(end.max()) + 2, 2):
    live = int(((start <= x) & (end > x)).sum())
    heavy = int(((start <= x) & (end > x) & (pop >= 4)).sum())
    print('t = %2d us: %5d waves resident (%4d with >= 4 primitives)' % (x, live, heavy))
