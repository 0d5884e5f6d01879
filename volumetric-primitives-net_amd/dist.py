"""Data-parallel sharding of the hot path: one process per GPU, batch split over ranks,
one RCCL all-reduce (sum) of the primitive-parameter gradients per step.

Every hot-path function is per-sample (sampler, Chamfer min, raster), so the only exchange
is the gradient (and loss) reduction.  The buffer is the GLOBAL [B_global, K, 10] gradient:
each rank fills its own slice, the rest stays zero, and the sum leaves the full gradient on
every rank.  The loss rides in the same buffer so a step is ONE collective."""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, rank, world):
    """Rank r of n takes samples [r*B/n, (r+1)*B/n).  Requires B % n == 0."""
    assert global_batch % world == 0, 'global batch must divide evenly over ranks'
    per = global_batch // world
    return rank * per, (rank + 1) * per


class GradAllReduce:
    """Persistent flat buffer [B_global*K*10 + 1]; `reduce(local_grad, local_loss)` returns
    (global grad [B_global,K,10], mean loss).  Backend 'nccl' is RCCL over xGMI on ROCm;
    'gloo' is used by the CPU tests."""

    def __init__(self, global_batch, K, device, rank=None, world=None):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.lo, self.hi = shard_bounds(global_batch, self.rank, self.world)
        self.shape = (global_batch, K, 10)
        self.buf = torch.zeros(global_batch * K * 10 + 1, dtype=torch.float32, device=device)
        self.grad = self.buf[:-1].view(self.shape)

    def reduce(self, local_grad, local_loss):
        self.buf.zero_()
        self.grad[self.lo:self.hi].copy_(local_grad)
        self.buf[-1] = local_loss.detach() / self.world
        if self.world > 1 or dist.is_initialized():
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
        return self.grad, self.buf[-1]
