"""modules/sampling of the reference on the HIP sampler kernel.

`Sampling.sphere_sampling(v, q, t, num_points)` / `cuboid_sampling(...)` keep the
reference signatures (sampling.py:11-37, call site train.py:117).  The packed entry
point `Sampling.sample_primitives` does the whole per-primitive loop + torch.cat of
sample_predict_points (train.py:105-120) in one launch."""
import torch

from ..ops import SampleFunction, SPHERE, CUBOID, kinds_tensor
from ..primitives import pack_primitives


class Sampling:
    # The reference draws its uniforms from torch's global generator (sphere.py:26-27,
    # cuboid.py:66).  Here the uniforms come from in-kernel Philox; the 63-bit key of each call
    # is drawn from torch's global (CPU) generator, so torch.manual_seed (train.py:25) makes a
    # run reproducible and successive calls get independent streams, like the reference.
    def __init__(self):
        pass

    @staticmethod
    def _next_stream():
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())

    @classmethod
    def _single(cls, kind, v, q, t, num_points, u, seed, sample_base):
        cls.check_parameters(v, q, t)
        assert type(num_points) == int and num_points > 0      # sphere.py:17-19
        params = torch.cat([v, q, t], 1)[:, None, :]
        kinds = kinds_tensor([kind], params.device)
        if u is not None:
            u = u.reshape(v.size(0), 1, num_points, 3)
        elif seed is None:
            seed = cls._next_stream()
        return SampleFunction.apply(params, kinds, u, seed or 0, sample_base, num_points)

    @classmethod
    def cuboid_sampling(cls, v, q, t, num_points: int = 1000, u=None, seed=None, sample_base=0):
        """sampling.py:11-23.  Optional `u` (B,N,3): explicit uniform draws (cuboid.py:66)."""
        return cls._single(CUBOID, v, q, t, num_points, u, seed, sample_base)

    @classmethod
    def sphere_sampling(cls, v, q, t, num_points: int = 1000, u=None, seed=None, sample_base=0):
        """sampling.py:25-37.  Optional `u` (B,N,3): u[...,0] elev draw, u[...,1] azim draw
        (sphere.py:26-27)."""
        return cls._single(SPHERE, v, q, t, num_points, u, seed, sample_base)

    @classmethod
    def cone_sampling(cls, v, q, t, num_points: int = 1000):
        """sampling.py:39-45 is `pass` in the reference: kept as the same stub."""
        pass

    @classmethod
    def sample_primitives(cls, params, kinds, num_points: int, u=None, seed=None, sample_base=0):
        """All K primitives of all B samples in one launch: params [B,K,10] (or the K lists of
        the network through pack_primitives) -> [B, K*num_points, 3], primitive-major like
        torch.cat(dim=1) at train.py:119."""
        if isinstance(params, (tuple, list)):
            params = pack_primitives(*params)
        kinds = kinds_tensor(kinds, params.device)
        if u is None and seed is None:
            seed = cls._next_stream()
        return SampleFunction.apply(params, kinds, u, seed or 0, sample_base, num_points)

    @staticmethod
    def check_parameters(v, q, t):
        B = v.size(0)                    # sampling.py:47-52
        assert v.size() == (B, 3)
        assert q.size() == (B, 4)
        assert t.size() == (B, 3)
