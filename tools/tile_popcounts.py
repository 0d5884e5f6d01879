"""Visible primitives per 16x16 tile at the C3 workload (from the tile masks the raster stores): distribution over the
tiles and per tile POSITION (mean over the batch), in the launch order (centre-out) of make_tile."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vpn_amd
from vpn_amd import _lib
from bench import synth_inputs
dev = torch.device('cuda')
B, K, H, W = 64, 32, 256, 256
params, _ = synth_inputs(B, K, 8, 1234, dev)
kinds = vpn_amd.kinds_tensor([0] * K, dev)
cam = torch.tensor([[1.0, 0.0, 0.0]], device=dev).expand(B, 3).contiguous()
L = _lib.lib()
rec = torch.zeros((L.vpn_raster_records_size(B, K, H, W) // 4,), dtype=torch.float32, device=dev)
lws = torch.zeros((L.vpn_raster_loss_workspace(B, H, W) // 4,), dtype=torch.float32, device=dev)
ws = torch.zeros((L.vpn_raster_bwd_workspace(B, K, H, W) // 4,), dtype=torch.float32, device=dev)
gt = torch.zeros(B, H, W, device=dev)
losses = torch.zeros(4, device=dev)
_lib.call('vpn_raster_total_fwd_fin', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, 0.05, 0.1, 2.0, _lib.ptr(gt), _lib.ptr(gt), 0,
          1.0, 1.0, _lib.ptr(rec), _lib.ptr(lws), _lib.ptr(ws), 0, None, 0, 0, 0, 0.0, 0.0, 0.0, _lib.ptr(losses), None, None, None, _lib.stream())
torch.cuda.synchronize()
ntile = 256
masks = rec.view(torch.int64)[L.vpn_raster_records_size(B, K, H, W) // 8 - B * ntile:].cpu()
pop = torch.tensor([bin(int(m) & ((1 << 64) - 1)).count('1') for m in masks.tolist()]).reshape(B, ntile)
print('tiles:', pop.numel(), 'mean visible primitives per tile %.2f' % pop.float().mean(), 'max', int(pop.max()))
h = torch.bincount(pop.flatten(), minlength=12)
print('histogram (visible primitives: tiles):', {i: int(c) for i, c in enumerate(h.tolist()) if c})
co = lambda i, n: (n // 2 - (i + 1) // 2) if (i & 1) else (n // 2 + (i + 1) // 2)
order = [co(iy, 16) * 16 + co(ix, 16) for iy in range(16) for ix in range(16)]
per_pos = pop.float().mean(0)
print('mean popcount per tile position in launch order (16 per line):')
for r in range(16):
    print(' '.join('%4.1f' % per_pos[order[r * 16 + c]] for c in range(16)))
