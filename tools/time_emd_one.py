"""One auction workload, a few launches (profiling target): python tools/time_emd_one.py [uniform|step|converged] [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd, bench
mode = sys.argv[1] if len(sys.argv) > 1 else 'step'
B, n = int(sys.argv[2]) if len(sys.argv) > 2 else 64, 2048
dev = torch.device('cuda')
if mode == 'uniform':
    g = torch.Generator().manual_seed(1)
    x1 = torch.rand(B, n, 3, generator=g).to(dev); x2 = torch.rand(B, n, 3, generator=g).to(dev)
elif mode == 'converged':       # bench.py's c5.fused_partly_converged: GT on a target's surfaces, predictions = the target perturbed by 10 %
    params, kinds, x2 = bench.c5_inputs(vpn_amd, B, 64, 32, 64, dev, 'surface')[:3]
    x1 = vpn_amd.Sampling.sample_primitives(params, kinds, 32, seed=1234)
else:
    params, x2 = bench.synth_inputs(B, 64, n, 1234, dev)
    x1 = vpn_amd.Sampling.sample_primitives(params, vpn_amd.kinds_tensor([0] * 64, dev), 32, seed=1234)
emd = vpn_amd.modules.loss.EarthMoverDistanceLoss()
for _ in range(2):
    d, a = emd(x1, x2, 0.005, 50)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    d, a = emd(x1, x2, 0.005, 50)
torch.cuda.synchronize()
print('%s B=%d: %.3f ms' % (mode, B, (time.perf_counter() - t) / 3 * 1e3), flush=True)
