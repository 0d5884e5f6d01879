"""Raster kernels alone at the C3 and C2 workloads: per-kernel times (library launch profiler) and HIP-graph replay
time of the one-pass step (bin + total + finalize + finish)."""
import sys, torch
sys.path.insert(0, '.')
import vpn_amd
from vpn_amd import _lib
from bench import synth_inputs
dev = torch.device('cuda')
for (B, K, H) in ((64, 32, 256), (32, 16, 128)):
    W = H
    params, gt = synth_inputs(B, K, 8, 1234, dev)
    p2, _ = synth_inputs(B, K, 8, 4321, dev)
    kinds = vpn_amd.kinds_tensor([0] * K, dev)
    cam = torch.tensor([[1.0, 0.0, 0.0]], device=dev).expand(B, 3).contiguous()
    with torch.no_grad():
        a2, d2 = vpn_amd.RasterFunction.apply(p2, kinds, cam, H, W, 0.05, 0.1, 2.0)
    gs, gd = (a2 > 0.5).float(), d2.clone()
    p = params.clone().requires_grad_(True)
    one = torch.ones((), device=dev)
    def step():
        p.grad = None
        out = vpn_amd.RasterTotalFunction.apply(p, kinds, cam, gs, gd, H, W, 0.05, 0.1, 2.0, False, 1.0, 1.0)
        out[2].backward(one)
    def step2():
        p.grad = None
        out = vpn_amd.RasterLossFunction.apply(p, kinds, cam, gs, gd, H, W, 0.05, 0.1, 2.0, False)
        out.sum().backward()
    for _ in range(3): step(); step2()
    with _lib.KernelProfile() as kp:
        for _ in range(20): step()
    r1 = {k: round(v[1] * 1e3, 1) for k, v in kp.summary().items()}
    with _lib.KernelProfile() as kp:
        for _ in range(20): step2()
    r2 = {k: round(v[1] * 1e3, 1) for k, v in kp.summary().items()}
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        step()
    for _ in range(10): g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): g.replay()
    b.record(); torch.cuda.synchronize()
    print('B=%d K=%d %dx%d  one-pass: %s sum %.1f us | graph replay %.1f us/step | two-call: %s' % (B, K, H, W, r1, sum(r1.values()), a.elapsed_time(b) * 1e3 / 200, r2))
