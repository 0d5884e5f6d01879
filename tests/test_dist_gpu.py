"""Two ranks on ONE GPU: the HIP hot path under world_size 2.  Each child process initialises the
device itself, runs HotPathLossFunction on its shard (Philox keyed by the global sample index: sample_base = first
sample of the shard), and the two gradient slices meet through vpn_amd.dist's reducers over gloo (CPU copies; RCCL
needs one GPU per rank).  The result must equal the single-process batch: bit for bit for the sampler + Chamfer
part, <= 1e-6 for the raster part.

The children are forked from a fork SERVER that tests/conftest.py starts before any test touches the GPU: no process
that has initialised the GPU ever forks or execs (the pool forbids the exec)."""
import multiprocessing as mp
import os
import sys

import pytest
import torch

from conftest import ROOT

B, K, n, M, H, W = 4, 4, 64, 256, 32, 32
SEED = 4242


def _inputs():
    g = torch.Generator().manual_seed(7)
    v = (torch.rand(B, K, 3, generator=g) + 0.1) / torch.tensor([8.0, 10.0, 10.0])
    params = torch.cat([v, torch.rand(B, K, 4, generator=g), 0.35 * (torch.rand(B, K, 3, generator=g) * 2 - 1)], 2)
    gt_pts = torch.rand(B, M, 3, generator=g) - 0.5
    gt_sil = (torch.rand(B, 1, H, W, generator=g) > 0.5).float()
    gt_dep = 2.0 - torch.rand(B, H, W, generator=g)
    cam = torch.tensor([[1.0, 0.0, 0.0]]).expand(B, 3).contiguous()
    return params, gt_pts, gt_sil, gt_dep, cam


def _step(lo, hi, weights):
    """Loss and gradient of samples [lo, hi) on the GPU: (local mean loss, grad [hi-lo, K, 10]) as CPU tensors."""
    import vpn_amd
    dev = torch.device('cuda', 0)
    params, gt_pts, gt_sil, gt_dep, cam = _inputs()
    p = params[lo:hi].to(dev).requires_grad_(True)
    out = vpn_amd.HotPathLossFunction.apply(p, vpn_amd.kinds_tensor([1, 0, 0, 0], dev), cam[lo:hi].to(dev), gt_pts[lo:hi].to(dev),
                                            gt_sil[lo:hi].to(dev), gt_dep[lo:hi].to(dev), n, SEED, lo, H, W, 0.05, 0.1, 2.0,
                                            *weights)
    out[2].backward()
    return out[2].detach().cpu(), p.grad.cpu()


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)                                   # both ranks share the one device of the box
    from vpn_amd.dist import GradAllGather, GradAllReduce, shard_bounds
    lo, hi = shard_bounds(B, rank, world)
    res = {}
    for name, weights in (('cd', (1.0, 0.0, 0.0)), ('all', (0.7, 1.3, 0.4))):
        loss, grad = _step(lo, hi, weights)
        g1, l1 = GradAllReduce(B, K, torch.device('cpu'), rank, world).reduce(grad, loss)
        g2, l2 = GradAllGather(B, K, torch.device('cpu'), rank, world).reduce(grad, loss)
        res[name] = {'ar_grad': g1.clone(), 'ar_loss': l1.clone(), 'ag_grad': g2.clone(), 'ag_loss': l2.clone()}
    if rank == 0:
        torch.save(res, os.path.join(outdir, 'r0.pt'))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    ctx = mp.get_context('forkserver')                         # server started by conftest before the GPU was touched
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, 'rank process failed (exit code %s)' % p.exitcode
    got = torch.load(os.path.join(str(tmp_path), 'r0.pt'), weights_only=True)
    for name, weights in (('cd', (1.0, 0.0, 0.0)), ('all', (0.7, 1.3, 0.4))):
        loss, grad = _step(0, B, weights)                      # the whole batch in this process
        for kind in ('ar', 'ag'):
            g, l = got[name][kind + '_grad'], got[name][kind + '_loss']
            if name == 'cd':                                   # sampler + Chamfer: bit for bit (B and world are powers of 2)
                assert torch.equal(g, grad), kind
                assert abs(float(l) - float(loss)) <= 1e-7 * abs(float(loss))
            else:
                assert float((g - grad).abs().max() / grad.abs().max()) <= 1e-6, kind
                assert abs(float(l) - float(loss)) <= 1e-6 * abs(float(loss))
