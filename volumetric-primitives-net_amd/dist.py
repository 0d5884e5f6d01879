"""Data-parallel sharding of the hot path: one process per GPU, batch split over ranks,
one RCCL all-reduce (sum) of the primitive-parameter gradients per step.

Every hot-path function is per-sample (sampler, Chamfer min, raster), so the only exchange
is the gradient (and loss) of disjoint samples.  Two equivalent collectives:
  * GradAllGather (what bench.py runs): every rank contributes its own [B_local, K, 10] slice plus its loss
    and an all-gather leaves the global gradient on every rank — half the bytes and half the ring steps of an
    all-reduce, nothing to zero, and the packing is a plain copy that can sit inside a captured HIP graph;
  * GradAllReduce: the GLOBAL [B_global, K, 10] buffer, own slice filled, rest zero, summed.
The loss rides in the same buffer so a step is ONE collective either way."""
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
    'gloo' is used by the CPU tests.  Same contract as GradAllGather: `local_grad` is the gradient of the rank's
    LOCAL mean loss; both scale it by 1/world so the result is the gradient of the global-batch mean.
    `pack` (capturable: only enqueues kernels on the current stream) / `allreduce` (the collective) / `views`."""

    def __init__(self, global_batch, K, device, rank=None, world=None):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.lo, self.hi = shard_bounds(global_batch, self.rank, self.world)
        self.shape = (global_batch, K, 10)
        self.buf = torch.zeros(global_batch * K * 10 + 1, dtype=torch.float32, device=device)
        self.grad = self.buf[:-1].view(self.shape)

    def pack(self, local_grad, local_loss):
        self.buf.zero_()
        torch.mul(local_grad, 1.0 / self.world, out=self.grad[self.lo:self.hi])
        torch.mul(local_loss.detach().reshape(1), 1.0 / self.world, out=self.buf[-1:])

    def allreduce(self):
        if dist.is_initialized():
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
        elif self.world > 1:
            raise RuntimeError('GradAllReduce with world=%d needs an initialised process group' % self.world)

    def views(self):
        return self.grad, self.buf[-1]

    def reduce(self, local_grad, local_loss):
        self.pack(local_grad, local_loss)
        self.allreduce()
        return self.views()


class GradAllGather:
    """Same result as GradAllReduce through one all-gather.  `pack(local_grad, local_loss)` only enqueues
    copies on the current stream (capturable); `gather()` is the collective; `views()` exposes the gathered buffer
    without a copy, `result()` returns (global grad [B_global, K, 10], mean loss).  `reduce()` = the three in a row."""

    def __init__(self, global_batch, K, device, rank=None, world=None):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.lo, self.hi = shard_bounds(global_batch, self.rank, self.world)
        self.per, self.K, self.global_batch = self.hi - self.lo, K, global_batch
        self.G = self.per * K * 10
        self.row = self.G + 4                                   # + loss, padded to 16 bytes
        self.local = torch.zeros(self.row, dtype=torch.float32, device=device)
        self.all = torch.zeros(self.world, self.row, dtype=torch.float32, device=device)

    def pack(self, local_grad, local_loss):
        # d(global mean loss) = d(local mean loss) / world for this rank's samples
        torch.mul(local_grad.reshape(-1), 1.0 / self.world, out=self.local[:self.G])
        self.local[self.G:self.G + 1].copy_(local_loss.detach().reshape(1))

    def gather(self):
        if dist.is_initialized():
            if dist.get_backend() == 'gloo':                    # the CPU tests
                dist.all_gather([self.all[r] for r in range(self.world)], self.local)
            else:
                dist.all_gather_into_tensor(self.all.view(-1), self.local)
        elif self.world > 1:
            raise RuntimeError('GradAllGather with world=%d needs an initialised process group' % self.world)
        else:
            self.all[0].copy_(self.local)

    def views(self):
        """No-copy views of the gathered buffer: grad [world, B_local, K, 10] (rank-major = global sample order)
        and the per-rank losses [world]."""
        return self.all[:, :self.G].view(self.world, self.per, self.K, 10), self.all[:, self.G]

    def result(self):
        grad, losses = self.views()
        return grad.reshape(self.global_batch, self.K, 10), losses.mean()

    def reduce(self, local_grad, local_loss):
        self.pack(local_grad, local_loss)
        self.gather()
        return self.result()
