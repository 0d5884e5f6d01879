"""Per-round timeline of the auction kernel (library built with VPN_EXTRA_FLAGS=-DEMD_TRACE):
    python tools/emd_timeline.py [uniform|step|converged]
for sample 0: per round the bidders, the team size, rows / targets evaluated, and where the round's time went (list build,
bids, waiting for the other workgroups' granules, assign), slowest workgroup of the sample."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd, bench
from vpn_amd import _lib
dev = torch.device('cuda')
mode = sys.argv[1] if len(sys.argv) > 1 else 'step'
B, n = int(os.environ.get('B', 64)), 2048
if mode == 'uniform':
    g = torch.Generator().manual_seed(1)
    x1 = torch.rand(B, n, 3, generator=g).to(dev); x2 = torch.rand(B, n, 3, generator=g).to(dev)
elif mode == 'converged':       # bench.py's c5.fused_partly_converged: GT on a target's surfaces, predictions = the target perturbed by 10 %
    params, kinds, x2 = bench.c5_inputs(vpn_amd, B, 64, 32, 64, dev, 'surface')[:3]
    x1 = vpn_amd.Sampling.sample_primitives(params, kinds, 32, seed=1234)
else:
    params, x2 = bench.synth_inputs(B, 64, n, 1234, dev)
    x1 = vpn_amd.Sampling.sample_primitives(params, vpn_amd.kinds_tensor([0] * 64, dev), 32, seed=1234)
L = _lib.lib()
ws = torch.zeros(L.vpn_emd_workspace(B, n) // 4, dtype=torch.int32, device=dev)
dist = torch.empty(B, n, device=dev); asg = torch.empty(B, n, dtype=torch.int32, device=dev)
for _ in range(3):
    _lib.call('vpn_emd_fwd', _lib.ptr(x1.contiguous()), _lib.ptr(x2.contiguous()), B, n, 0.005, 50, _lib.ptr(dist), _lib.ptr(asg), _lib.ptr(ws), int(os.environ.get('G', 0)), _lib.stream())
torch.cuda.synchronize()
w = ws.cpu().numpy().view('uint32')
for b in (0, B // 2):
    tr = w[b * 10 * n + 4 * n: b * 10 * n + 4 * n + 16 * 64 * 8].reshape(16, 64, 8).astype('int64')
    G = int((tr[:, 0, 0] != 0).sum())
    t_launch = tr[:G, 63, 7].min()
    print('sample %d: G = %d; first round starts %.1f us after the kernel began; last round ends at %.1f us' % (b, G, (tr[:G, 0, 0].min() - t_launch) / 100.0, (tr[:G, :50, 4].max() - t_launch) / 100.0))
    ft = w[b * 10 * n + 8 * n: b * 10 * n + 8 * n + 16 * 64 * 4].reshape(16, 64, 4).astype('int64')   # balanced form: stamps after A, B, C; rows + batches << 24
    print(' it     U  Uown(max)  T  rows/bid evals/bid | list  bid(max) bid(min)  wait(min)  assign | round us | balanced form, slowest workgroup: A  B  C  D us, rows, batches')
    for it in range(50):
        r = tr[:G, it]
        if r[0, 0] == 0: break
        U = r[0, 5] & 0xffff; uown = (r[:, 5] >> 16)
        T = 1
        while T < int(os.environ.get('VPN_EMD_TMAX', 16)) and 2 * T * int(uown.max()) <= int(os.environ.get('VPN_EMD_TNUM', 1024)): T *= 2
        us = lambda a: a / 100.0
        print(' %2d  %4d  %4d      %2d  %7.1f %8.1f | %4.1f  %6.1f  %6.1f   %6.1f   %5.1f | %6.1f' % (
            it, U, uown.max(), T, r[:, 7].sum() / max(1, uown.sum()), r[:, 6].sum() / max(1, uown.sum()),
            us((r[:, 1] - r[:, 0]).max()), us((r[:, 2] - r[:, 1]).max()), us((r[:, 2] - r[:, 1]).min()), us((r[:, 3] - r[:, 2]).min()),
            us((r[:, 4] - r[:, 3]).max()), us(r[:, 4].max() - r[:, 0].min())), end='')
        f = ft[:G, it]
        if f[:, 3].any():
            gs = int((r[:, 2] - r[:, 1]).argmax())
            print(' | %5.1f %5.1f %5.1f %5.1f  %5d %d' % (us(f[gs, 0] - r[gs, 1]), us(f[gs, 1] - f[gs, 0]), us(f[gs, 2] - f[gs, 1]), us(r[gs, 2] - f[gs, 2]),
                                                         f[gs, 3] & 0xffff, f[gs, 3] >> 24), ' ~%dk targets' % ((f[gs, 3] >> 16) & 0xff), ' C of wave 0: loop %d, last offers %d cycles' % (r[gs, 6], r[gs, 7]))
        else:
            print()
