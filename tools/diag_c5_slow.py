"""Why is the c5 block slower inside the default bench run than alone?  In a FRESH process per variant, one ingredient of
bench.py's C3 section runs first, then the fused C5 step is timed (GPU box): python tools/diag_c5_slow.py <variant>
variants: none | eager | graph | graph_noreplay | timer | profile | rasterfn"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, vpn_amd
from vpn_amd import _lib
dev = torch.device('cuda')
_lib.lib()
variant = sys.argv[1] if len(sys.argv) > 1 else 'none'

def c5(tag):
    r = bench.train_step_block(vpn_amd, _lib, dev, 64, 64, 32, 256, 30, 5, 3, 'fused', False)[0]
    print('%-40s c5 fused %.3f ms   %s' % (tag, r['ms_per_step'], {k: v['avg_us'] for k, v in list(r['kernel_us'].items())[:3]}), flush=True)

B, K, n, M, H = 64, 32, 256, 2048, 256
kinds = vpn_amd.kinds_tensor([0] * K, dev)
params_all, gt = bench.synth_inputs(B, K, M, 1234, dev)
params = params_all.clone().requires_grad_(True)
cam = torch.tensor([[1.0, 0.0, 0.0]], device=dev).expand(B, 3).contiguous()
p2, _ = bench.synth_inputs(B, K, M, 4321, dev)
if variant != 'none':
    with torch.no_grad():
        a2, d2 = vpn_amd.RasterFunction.apply(p2, kinds, cam, H, H, 0.05, 0.1, 2.0)
    gs, gd = (a2 > 0.5).float(), d2.clone()
one = torch.ones((), device=dev)
seed = torch.full((1,), 1234, dtype=torch.int64, device=dev)
def compute(_i=0):
    params.grad = None
    o = vpn_amd.HotPathLossFunction.apply(params, kinds, cam, gt, gs, gd, n, seed, 0, H, H, 0.05, 0.1, 2.0, 1.0, 1.0, 1.0, 1.0, 1.0, False, True)
    o[2].backward(one)
    return o[2]
if variant == 'eager':
    for i in range(20): compute(i)
elif variant in ('graph', 'graph_noreplay'):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3): compute(i)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        compute(0)
    if variant == 'graph':
        for i in range(400): g.replay()
elif variant == 'timer':
    with _lib.KernelTimer() as kt:
        for i in range(20): compute(i)
        kt.summary()
elif variant == 'profile':
    with _lib.KernelProfile() as kp:
        for i in range(20): compute(i)
    kp.summary()
torch.cuda.synchronize()
c5('first c5 after: ' + variant)
c5('second c5')
