"""Generate golden vectors by running the REFERENCE's own pure-PyTorch leaf files.

Runs only in the authoring container (needs /root/reference); the GPU box never
sees the reference, only the .npz fixtures this script writes to tests/golden/.
Loader = SURVEY.md Appendix A: config.DEVICE is flipped to 'cpu' before import,
`modules` / `modules.loss` are registered as empty namespace packages so their
__init__ files (which need kaolin / torchvision / the compiled emd extension) do
not execute, and `device_copy_semantics()` restores on CPU the differentiable
copy that `.to('cuda')` performs at modules/transform/rotate.py:34.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def load_reference():
    sys.path.insert(0, REF)
    import config
    config.DEVICE = 'cpu'
    for name, path in [('modules', REF + '/modules'), ('modules.loss', REF + '/modules/loss')]:
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    from modules.sampling import Sampling
    from modules.loss.chamfer_distance import ChamferDistanceLoss
    from modules.loss.vp_diverse import VPDiverseLoss
    from modules.transform import (transform_points, view_to_obj_points, obj_to_view_points,
                                   rotate_points, rotate_points_forward_x_axis)
    return dict(Sampling=Sampling, Chamfer=ChamferDistanceLoss, VPDiverse=VPDiverseLoss,
                transform_points=transform_points, view_to_obj_points=view_to_obj_points,
                obj_to_view_points=obj_to_view_points, rotate_points=rotate_points,
                rotate_points_forward_x_axis=rotate_points_forward_x_axis, config=config)


@contextlib.contextmanager
def device_copy_semantics():
    orig = torch.Tensor.to

    def to(self, *a, **k):
        out = orig(self, *a, **k)
        return self.clone() if (out is self and self.is_leaf and self.requires_grad) else out
    torch.Tensor.to = to
    try:
        yield
    finally:
        torch.Tensor.to = orig


def rand_params(g, B, extreme=False):
    v = (torch.rand(B, 3, generator=g) + 0.1) / torch.tensor([8.0, 10.0, 10.0])
    if extreme:
        v[0] = torch.tensor([0.5, 0.01, 0.02])        # extreme aspect ratio for the face quotas
    q = torch.rand(B, 4, generator=g)
    q[:, :3] = q[:, :3] * 2 - 1                        # axis is not pre-normalised in the reference
    q[-1, 3] = 1.37                                    # exercises the `% 1` of rotate.py:63
    t = 0.35 * (torch.rand(B, 3, generator=g) * 2 - 1)
    return v, q, t


def npz(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    out = {}
    for k, a in arrs.items():
        out[k] = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print('wrote', name, {k: tuple(v.shape) for k, v in out.items()})


def sampler_case(ref, kind, B, N, seed, extreme=False):
    g = torch.Generator().manual_seed(seed)
    v, q, t = rand_params(g, B, extreme)
    W = torch.randn(B, N, 3, generator=g)
    v.requires_grad_(True); q.requires_grad_(True); t.requires_grad_(True)
    fn = ref['Sampling'].sphere_sampling if kind == 'sphere' else ref['Sampling'].cuboid_sampling
    torch.manual_seed(seed)
    with device_copy_semantics():
        pts = fn(v, q, t, N)
        (pts * W).sum().backward()
    # replay the draws: sphere.py:26-27 (elev first, azim second) / cuboid.py:66
    torch.manual_seed(seed)
    if kind == 'sphere':
        u1 = torch.rand((B, N, 1)); u2 = torch.rand((B, N, 1))
        u = torch.cat([u1, u2, torch.zeros_like(u1)], 2)
    else:
        u = torch.rand((B, N, 3))
    return dict(v=v, q=q, t=t, u=u, W=W, points=pts, grad_v=v.grad, grad_q=q.grad, grad_t=t.grad)


def main():
    ref = load_reference()
    torch.set_num_threads(1)

    # G1 sphere sampler
    npz('g1_sphere_b4_n128', **sampler_case(ref, 'sphere', 4, 128, 11))
    npz('g1_sphere_b2_n7', **sampler_case(ref, 'sphere', 2, 7, 12))
    # G2 cuboid sampler (one extreme aspect ratio); face counts from the reference helper
    c = sampler_case(ref, 'cuboid', 3, 128, 21, extreme=True)
    from modules.sampling.cuboid import get_faces_points
    vv = c['v'].detach()
    c['counts'] = get_faces_points(vv[:, 0:1], vv[:, 1:2], vv[:, 2:3], 128)
    npz('g2_cuboid_b3_n128', **c)

    # G3 multi-primitive order of train.py:105-120: cuboids first, then spheres
    import importlib.util
    B, n, K = 2, 16, 3
    g = torch.Generator().manual_seed(31)
    vs, qs, ts = [], [], []
    for k in range(K):
        v, q, t = rand_params(g, B)
        vs.append(v); qs.append(q); ts.append(t)
    torch.manual_seed(1234)
    funcs = [ref['Sampling'].cuboid_sampling, ref['Sampling'].sphere_sampling, ref['Sampling'].sphere_sampling]
    with device_copy_semantics():
        pts = torch.cat([funcs[k](vs[k], qs[k], ts[k], n) for k in range(K)], 1)   # train.py:112-119
    torch.manual_seed(1234)
    us = [torch.rand((B, n, 3))]
    for k in (1, 2):
        u1 = torch.rand((B, n, 1)); u2 = torch.rand((B, n, 1))
        us.append(torch.cat([u1, u2, torch.zeros_like(u1)], 2))
    params = torch.stack([torch.cat([vs[k], qs[k], ts[k]], 1) for k in range(K)], 1)
    npz('g3_multi_b2_k3_n16', params=params, types=np.array([1, 0, 0], dtype=np.int32),
        u=torch.stack(us, 1), points=pts)

    # G4 Chamfer: loss, argmin both directions, grads
    cd = ref['Chamfer']()
    for (B, N, M, seed) in [(4, 128, 96, 41), (2, 257, 2048, 42)]:
        g = torch.Generator().manual_seed(seed)
        p1 = (torch.rand(B, N, 3, generator=g) - 0.5).requires_grad_(True)
        p2 = (torch.rand(B, M, 3, generator=g) - 0.5).requires_grad_(True)
        loss = cd(p1, p2)
        loss.backward()
        loss_b = cd(p1.detach(), p2.detach(), each_batch=True, w1=0.5, w2=2.0)
        with torch.no_grad():
            diff = p1[:, :, None, :] - p2[:, None, :, :]
            dist = torch.sum(diff * diff, dim=3)
            m1, i1 = torch.min(torch.sqrt(dist), dim=2)
            m2, i2 = torch.min(torch.sqrt(torch.transpose(dist, 1, 2)), dim=2)
        npz('g4_chamfer_b%d_n%d_m%d' % (B, N, M), p1=p1, p2=p2, loss=loss, loss_each_w=loss_b,
            min1=m1, idx1=i1.int(), min2=m2, idx2=i2.int(), grad_p1=p1.grad, grad_p2=p2.grad)
    # tie case (duplicated targets -> first index wins) and coincident-point case (NaN grad)
    p1 = torch.tensor([[[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.25, 0.5, 0.0]]], requires_grad=True)
    p2 = torch.tensor([[[0.0, 1.0, 0.0], [0.0, -1.0, 0.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [2.0, 0.0, 0.0]]],
                      requires_grad=True)
    loss = cd(p1, p2)
    loss.backward()
    with torch.no_grad():
        diff = p1[:, :, None, :] - p2[:, None, :, :]
        dist = torch.sum(diff * diff, dim=3)
        m1, i1 = torch.min(torch.sqrt(dist), dim=2)
        m2, i2 = torch.min(torch.sqrt(torch.transpose(dist, 1, 2)), dim=2)
    npz('g4_chamfer_ties', p1=p1, p2=p2, loss=loss, min1=m1, idx1=i1.int(), min2=m2, idx2=i2.int(),
        grad_p1=p1.grad, grad_p2=p2.grad)

    # G5 VPDiverse (K = config.VP_NUM = 16 centres vs M = 64)
    g = torch.Generator().manual_seed(51)
    B, K, M = 3, ref['config'].VP_NUM, 64
    ts = [(0.35 * (torch.rand(B, 3, generator=g) * 2 - 1)).requires_grad_(True) for _ in range(K)]
    gt = torch.rand(B, M, 3, generator=g) - 0.5
    loss = ref['VPDiverse']()(ts, gt)
    loss.backward()
    npz('g5_vpdiverse_b3_k16_m64', translates=torch.stack([t.detach() for t in ts], 1), gt=gt, loss=loss,
        grad_t=torch.stack([t.grad for t in ts], 1))

    # G6 view <-> object transforms and transform_points
    g = torch.Generator().manual_seed(61)
    B, N = 3, 50
    pts = torch.rand(B, N, 3, generator=g) - 0.5
    dists = 1 + torch.rand(B, generator=g)
    elevs = 20 + 20 * torch.rand(B, generator=g)
    azims = 360 * torch.rand(B, generator=g)
    angles = 360 * torch.rand(B, generator=g)
    with device_copy_semantics():
        o2v = ref['obj_to_view_points'](pts, dists, elevs, azims)
        v2o = ref['view_to_obj_points'](pts, dists, elevs, azims, angles)
        rx = ref['rotate_points_forward_x_axis'](pts, angles)
    _, q, t = rand_params(g, B)
    W = torch.randn(B, N, 3, generator=g)
    pr = pts.clone().requires_grad_(True); q.requires_grad_(True); t.requires_grad_(True)
    with device_copy_semantics():
        tp = ref['transform_points'](pr, q, t)
        (tp * W).sum().backward()
    npz('g6_transforms', points=pts, dists=dists, elevs=elevs, azims=azims, angles=angles,
        obj_to_view=o2v, view_to_obj=v2o, rot_x=rx, q=q, t=t, W=W, transform=tp,
        grad_points=pr.grad, grad_q=q.grad, grad_t=t.grad)


if __name__ == '__main__':
    main()
