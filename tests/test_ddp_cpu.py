"""Row f4 remainder: the DDP training loop of examples/train_ddp.py (reference: train.py:211-296,
modules/network/vpnet_one_resnet.py:28-43) over gloo with world_size 2 on the CPU.  The HIP operators need a GPU, so
the loss injected here is the oracle's restatement of the same step (head post-processing -> sampler with Philox
draws keyed by the global sample index -> Chamfer + silhouette loss); what is under test is the loop: sharding,
sample_base, DDP with find_unused_parameters, bucketed all-reduce — two ranks must end with the weights a single
process gets on the whole batch."""
import importlib.util
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

K, SAMPLE_NUM, SIZE, STEPS, GLOBAL_BATCH = 4, 12, 16, 3, 4


def _example():
    spec = importlib.util.spec_from_file_location('train_ddp', os.path.join(ROOT, 'examples', 'train_ddp.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def oracle_loss(heads_out, batch, kinds, sample_num, seed, sample_base, size):
    from oracle import vpn_oracle as O
    gt_points, gt_sil, cam = batch
    params = O.head_post_process(*heads_out)
    u = O.philox_uniforms(seed, sample_base, params.shape[0], len(kinds), sample_num)
    pts = O.sample_primitives(params, kinds, u)
    alpha, _ = O.raster(params, kinds, cam, size, size, 0.05, 0.1, 2.0)
    return O.chamfer_loss(pts, gt_points) + O.silhouette_loss(alpha, gt_sil)


def _run(rank, world):
    ex = _example()
    return ex.run(rank, world, torch.device('cpu'), oracle_loss, steps=STEPS, global_batch=GLOBAL_BATCH, K=K, feat=16,
                  sample_num=SAMPLE_NUM, M=32, size=SIZE, bucket_cap_mb=0.01,      # tiny buckets: several all-reduces per step
                  make_optimizer=lambda p: torch.optim.SGD(p, lr=0.05))           # SGD: weights are linear in the gradients
                                                                                  # (Adam turns 1e-7 gradient noise into lr-sized steps)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    net = _run(rank, world)
    if rank == 0:
        torch.save({k: v.clone() for k, v in net.state_dict().items()}, out)
    dist.destroy_process_group()


def test_ddp_two_ranks_match_single_process(tmp_path):
    out = str(tmp_path / 'w.pt')
    port = 29900 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    ref = _run(0, 1).state_dict()
    assert set(got) == set(ref)
    for k in ref:
        if k.startswith('unused_fc'):
            assert torch.equal(got[k], ref[k])                 # never touched: no gradient, no update (find_unused_parameters)
        else:
            assert torch.allclose(got[k], ref[k], rtol=1e-5, atol=1e-6), k
    moved = sum(float((ref[k] - _example().Heads(16, K).state_dict()[k]).abs().sum()) for k in ref if 'trunk' in k)
    assert moved > 0                                           # the loop really trained something
