"""Would the Chamfer scan and the raster of the C3 step gain from running side by side (two HIP streams, fork / join
captured in a graph) instead of back to back?  Both launches as the step issues them (mode-7 scan without the rider,
raster with a tile order computed beforehand and without the Chamfer term), replayed as graphs: serial, forked, and each
one alone."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd
from vpn_amd import _lib
from bench import synth_inputs
dev = torch.device('cuda')
B, K, n, M, H, W = int(os.environ.get('B', 64)), 32, 256, 2048, 256, 256
N = K * n
params, gt_points = synth_inputs(B, K, M, 1234, dev)
p2, _ = synth_inputs(B, K, 8, 4321, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
cam = torch.tensor([[1.0, 0.0, 0.0]], device=dev).expand(B, 3).contiguous()
with torch.no_grad():
    a2, d2g = vpn_amd.RasterFunction.apply(p2, kinds, cam, H, W, 0.05, 0.1, 2.0)
gt_sil, gt_dep = (a2 > 0.5).float().contiguous(), d2g.contiguous()
L = _lib.lib()
f32 = lambda nbytes: torch.zeros((nbytes // 4,), dtype=torch.float32, device=dev)
rec, lws, rws = f32(L.vpn_raster_records_size(B, K, H, W)), f32(L.vpn_raster_loss_workspace(B, H, W)), f32(L.vpn_raster_bwd_workspace(B, K, H, W))
cws = f32(L.vpn_chamfer_workspace(B, N, M))
points = torch.empty(B, N, 3, device=dev)
d1, d2 = torch.empty(B, N, device=dev), torch.empty(B, M, device=dev)
i1, i2 = torch.empty(B, N, dtype=torch.int32, device=dev), torch.empty(B, M, dtype=torch.int32, device=dev)
order = torch.empty((L.vpn_raster_order_size(B, H, W) // 8,), dtype=torch.int64, device=dev)
losses = torch.zeros(4, device=dev)
P = _lib.ptr


def sp(stream):
    import ctypes
    return ctypes.c_void_p(stream.cuda_stream)


def sampler(s):
    _lib.call('vpn_hotpath_sample_fwd', P(params), P(kinds), None, 1234, None, 0, B, K, n, P(points), P(cam), H, W, 0.05, P(rec), P(lws),
              P(gt_points), M, P(cws), cws.numel() * 4, sp(s))


def scan_rider(s):
    _lib.call('vpn_hotpath_chamfer_fwd', P(points), P(gt_points), B, N, M, P(d1), P(i1), P(d2), P(i2), P(cws), cws.numel() * 4, 7, P(rec), K, H, W,
              P(order), sp(s))


def scan(s):
    _lib.call('vpn_chamfer_fwd_ws', P(points), P(gt_points), B, N, M, P(d1), P(i1), P(d2), P(i2), P(cws), cws.numel() * 4, 7, sp(s))


def raster(s):
    _lib.call('vpn_raster_total_fwd_fin', P(params), P(kinds), P(cam), B, K, H, W, 0.05, 0.1, 2.0, P(gt_sil), P(gt_dep), 0, 1.0, 1.0, P(rec), P(lws),
              P(rws), 1, None, 0, 0, 0, 0.0, 0.0, 0.0, P(losses), None, None, P(order), sp(s))


main = torch.cuda.Stream()
side = torch.cuda.Stream()
with torch.cuda.stream(main):
    sampler(main); scan_rider(main); raster(main); scan(main)
torch.cuda.synchronize()
ref = (losses.clone(), d1.clone(), i2.clone())


def graph_of(body):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        body()
    return g


def serial():
    scan(main); raster(main)


def forked():
    side.wait_stream(main)
    raster(side)
    scan(main)
    main.wait_stream(side)


def forked_raster_first():
    side.wait_stream(main)
    scan(side)
    raster(main)
    main.wait_stream(side)


def timed(g, reps=200, windows=5):
    out = []
    with torch.cuda.stream(main):
        for _ in range(20):
            g.replay()
        for _ in range(windows):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main)
            for _ in range(reps):
                g.replay()
            b.record(main)
            b.synchronize()
            out.append(a.elapsed_time(b) / reps * 1e3)
    return sorted(out)[len(out) // 2]


for name, body in (('scan alone', lambda: scan(main)), ('raster alone', lambda: raster(main)), ('serial scan -> raster', serial),
                   ('forked scan || raster', forked), ('forked, raster on the capturing stream', forked_raster_first)):
    g = graph_of(body)
    t = timed(g)
    torch.cuda.synchronize()
    ok = torch.equal(losses, ref[0]) and torch.equal(d1, ref[1]) and torch.equal(i2, ref[2])
    print('%-45s %7.1f us   results unchanged: %s' % (name, t, ok))


def eager(body, reps=200):
    with torch.cuda.stream(main):
        for _ in range(10):
            body()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(main)
        for _ in range(reps):
            body()
        b.record(main)
        b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print('eager serial %.1f us, eager forked %.1f us' % (eager(serial), eager(forked)))
