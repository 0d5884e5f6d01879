"""world_size-2 gloo test of the data-parallel path: batch sharded over ranks,
one all-reduce of the primitive-parameter gradients, result equal to the single-process batch.
The per-shard compute here is the CPU oracle (no GPU in this test); the sharding / reduction
logic under test is volumetric-primitives-net_amd/dist.py, the code bench.py runs with RCCL."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _loss_and_grad(params, gt, kinds, u):
    from oracle import vpn_oracle as O
    p = params.clone().requires_grad_(True)
    pts = O.sample_primitives(p, kinds, u)
    loss = O.chamfer_loss(pts, gt)
    loss.backward()
    return loss.detach(), p.grad


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import vpn_amd
    from vpn_amd.dist import GradAllGather, GradAllReduce, shard_bounds
    from oracle import vpn_oracle as O
    Bg, K, n, M = 4, 3, 16, 32
    g = torch.Generator().manual_seed(1234)
    params = torch.rand(Bg, K, 10, generator=g) * 0.3 + 0.05
    gt = torch.rand(Bg, M, 3, generator=g) - 0.5
    kinds = [1, 0, 0]
    u = O.philox_uniforms(1234, 0, Bg, K, n)              # keyed by GLOBAL sample index
    lo, hi = shard_bounds(Bg, rank, world)
    u_local = O.philox_uniforms(1234, lo, hi - lo, K, n)  # what a rank generates for its shard
    assert torch.equal(u_local, u[lo:hi])
    loss, grad = _loss_and_grad(params[lo:hi], gt[lo:hi], kinds, u_local)
    red = GradAllReduce(Bg, K, torch.device('cpu'), rank, world)
    # both reducers take the gradient of the LOCAL mean loss and scale by 1/world themselves (same contract)
    ggrad, gloss = red.reduce(grad, loss)
    # the all-gather formulation must give the same global gradient and loss from the same inputs
    gat = GradAllGather(Bg, K, torch.device('cpu'), rank, world)
    ggrad2, gloss2 = gat.reduce(grad, loss)
    assert torch.allclose(ggrad2, ggrad, rtol=1e-6, atol=1e-9) and torch.allclose(gloss2, gloss, rtol=1e-6)
    if rank == 0:
        torch.save({'grad': ggrad.clone(), 'loss': gloss.clone(), 'grad2': ggrad2.clone(), 'loss2': gloss2.clone()}, out)
    dist.destroy_process_group()


def test_dp_two_ranks_matches_single_process(tmp_path):
    from oracle import vpn_oracle as O
    out = str(tmp_path / 'r0.pt')
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    Bg, K, n, M = 4, 3, 16, 32
    g = torch.Generator().manual_seed(1234)
    params = torch.rand(Bg, K, 10, generator=g) * 0.3 + 0.05
    gt = torch.rand(Bg, M, 3, generator=g) - 0.5
    loss, grad = _loss_and_grad(params, gt, [1, 0, 0], O.philox_uniforms(1234, 0, Bg, K, n))
    assert torch.allclose(got['grad'], grad, rtol=1e-5, atol=1e-8)
    assert torch.allclose(got['loss'], loss, rtol=1e-5)
    assert torch.allclose(got['grad2'], grad, rtol=1e-5, atol=1e-8)
    assert torch.allclose(got['loss2'], loss, rtol=1e-5)


def test_shard_bounds():
    from vpn_amd.dist import shard_bounds
    assert [shard_bounds(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    import pytest
    with pytest.raises(AssertionError):
        shard_bounds(10, 0, 4)
