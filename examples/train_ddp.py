"""Data-parallel training loop of the reference (train.py:211-296) on the drop-in surface: BASELINE config C5
(K=64 primitives, 256x256, one process per GPU, DDP).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 examples/train_ddp.py [--steps 20]

The reference's ResNet-18 trunk + three heads (modules/network/vpnet_one_resnet.py:28-43, :100-106; 23.4 M
parameters = 93.5 MB of fp32 gradients per step at K=64) is out of scope: a stand-in
network with the same three heads takes its place, wrapped in torch's DistributedDataParallel over RCCL exactly as the
real one would be:

  * the whole loss of the step is ONE autograd node (HotPathLossFunction: sampler -> Chamfer -> raster + image losses),
    so backward runs  hot-path backward -> head post-processing backward -> network backward;  DDP's reducer launches
    the bucketed all-reduce of a bucket of NETWORK gradients (bucket_cap_mb, 25 MB default: 4 buckets for 93.5 MB) as
    soon as the bucket is complete, on its own stream: the collective of the last layers' buckets overlaps the
    backward of the earlier layers.  Nothing of the hot path is exchanged: its gradient w.r.t. (v, q, t) is per sample
    and flows into the local network replica (the kernel-only exchange of BASELINE config C4 is vpn_amd.dist);
  * the reference's resnet18 carries an `fc` layer its forward never uses (vpnet_one_resnet.py:45-57): under DDP that
    needs find_unused_parameters=True (or deleting the layer).  The stand-in has an unused head on purpose so the flag
    is exercised;
  * every rank draws its own surface points: the Philox key is (seed of the step, GLOBAL sample index), so the
    union over ranks is the batch a single process would have drawn (sample_base = rank * B_local);
  * the reference's loop divides by BATCH_SIZE constants and crashes on a last partial batch (train.py:127,137,289);
    nothing here depends on the batch size.

`run(rank, world, device, loss_fn, ...)` is the loop with the loss injectable: tests/test_ddp_cpu.py drives it over gloo on
the CPU with the oracle as the loss (the HIP operators need a GPU) and checks the DDP gradients against a single
process."""
import argparse
import os
import sys

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Heads(nn.Module):
    """Stand-in for VPNetOneRes (vpnet_one_resnet.py:28-43): features -> raw head outputs (volumes [B,3K], rotates
    [B,4K], translates [B,3K]).  `unused_fc` plays the resnet's never-called fc layer."""

    def __init__(self, feat, K, hidden=256):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(feat, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU())
        self.volume_fc, self.rotate_fc, self.translate_fc = nn.Linear(hidden, 3 * K), nn.Linear(hidden, 4 * K), nn.Linear(hidden, 3 * K)
        self.unused_fc = nn.Linear(hidden, 10)

    def forward(self, x):
        h = self.trunk(x)
        return self.volume_fc(h), self.rotate_fc(h), self.translate_fc(h)


def hip_loss(heads_out, batch, kinds, sample_num, seed, sample_base, size):
    """Loss of one step on the HIP hot path: head post-processing -> one autograd node for everything else."""
    import vpn_amd
    gt_points, gt_sil, cam = batch
    params = vpn_amd.pack_head_outputs(*heads_out)                                  # vpnet_one_resnet.py:34-41, :67-85
    out = vpn_amd.HotPathLossFunction.apply(params, vpn_amd.kinds_tensor(kinds, params.device), cam, gt_points, gt_sil, None,
                                            sample_num, seed, sample_base, size, size, vpn_amd.config.RASTER_SIGMA,
                                            vpn_amd.config.RASTER_GAMMA, vpn_amd.config.RASTER_Z_FAR, 1.0, 1.0, 0.0)
    return out[2]


def make_batch(B, K, feat, M, size, device, seed, lo, hi):
    """Synthetic global batch (seeded identically on every rank), this rank's slice [lo, hi)."""
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, feat, generator=g)
    gt_points = torch.rand(B, M, 3, generator=g) - 0.5
    gt_sil = (torch.rand(B, 1, size, size, generator=g) > 0.7).float()
    cam = torch.tensor([[1.0, 0.0, 0.0]]).expand(B, 3).contiguous()                 # train.py:172-174
    return feats[lo:hi].to(device), (gt_points[lo:hi].to(device), gt_sil[lo:hi].to(device), cam[lo:hi].to(device))


def run(rank, world, device, loss_fn, steps=3, global_batch=8, K=8, feat=32, sample_num=16, M=64, size=16, lr=1e-2,
        bucket_cap_mb=25, log=None, make_optimizer=None):
    """The DDP loop.  loss_fn(heads_out, batch, kinds, sample_num, seed, sample_base, size) -> scalar local-mean loss.
    Returns the model (unwrapped) after `steps` optimiser steps."""
    assert global_batch % world == 0
    per = global_batch // world
    lo, hi = rank * per, (rank + 1) * per
    torch.manual_seed(1234)                                                         # same initial weights on every rank
    net = Heads(feat, K).to(device)
    model = DDP(net, device_ids=[device.index] if device.type == 'cuda' else None, bucket_cap_mb=bucket_cap_mb,
                find_unused_parameters=True) if world > 1 or dist.is_initialized() else net
    opt = make_optimizer(net.parameters()) if make_optimizer else torch.optim.Adam(net.parameters(), lr=lr)   # train.py:93-103
    kinds = [0] * K                                                                 # config.py:33-34
    for it in range(steps):
        feats, batch = make_batch(global_batch, K, feat, M, size, device, 100 + it, lo, hi)
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(model(feats), batch, kinds, sample_num, 1000 + it, lo, size)  # local mean; DDP averages the gradients
        loss.backward()                                                             # bucketed all-reduce overlaps this
        opt.step()
        if log is not None:
            log(it, loss)
    return net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--global-batch', type=int, default=64)
    ap.add_argument('--prims', type=int, default=64)          # C5
    ap.add_argument('--sample-num', type=int, default=128)    # config.py:8
    ap.add_argument('--size', type=int, default=256)
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)    # RCCL over xGMI

    def log(it, loss):
        if rank == 0:
            print('step %3d  local loss %.5f' % (it, float(loss.detach())), flush=True)
    run(rank, world, dev, hip_loss, steps=args.steps, global_batch=args.global_batch, K=args.prims, feat=64,
        sample_num=args.sample_num, M=2048, size=args.size, lr=1e-3, log=log)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
