"""torch.autograd.Function wrappers over the C ABI (include/vpn_hip.h).

Pattern follows the reference's own native op, emdFunction (modules/loss/emd/
emd_module.py:29-70): forward allocates the outputs, calls native code, saves what
backward needs; backward returns one gradient per tensor input and None for the rest.
Unlike it, a non-zero return code raises."""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib

SPHERE, CUBOID = 0, 1
PARAM_STRIDE = 10


def _f32c(t):
    if not t.is_cuda:
        raise RuntimeError('vpn_amd operators run on the GPU only (got a %s tensor); there is no CPU path'
                           % t.device.type)
    return t.contiguous().float()          # emd_module.py:41-42 does the same to its inputs


import weakref

# Host copies of device kind tensors: id(tensor) -> (weak reference to it, the version counter it had then, the kinds as a
# tuple).  The entry dies with its tensor (weakref callback) and a hit also requires the reference to be the same object, so
# a later tensor that reuses the id or the device address is never mistaken for a known one.  Kind tensors made from a
# host list are cached per (kinds, device): the list train.py:112-116 implies is the same every step, and the B composed
# meshes of a batch share ONE tensor -- equality checks and validation then never touch the device.
_KINDS_HOST = {}
_KINDS_BY_TUPLE = {}


def _check_kinds(kinds):
    if any(k not in (SPHERE, CUBOID) for k in kinds):
        raise ValueError('unknown primitive kind in %r (0 = sphere, 1 = cuboid; cones are not implemented '
                         'in the reference either)' % (kinds,))


def _kinds_remember(t, tup):
    key = id(t)
    _KINDS_HOST[key] = (weakref.ref(t, lambda _r, key=key: _KINDS_HOST.pop(key, None)), t._version, tup)


def kinds_host(t):
    """The kinds of a device kind tensor as a host tuple: from the registry, or -- the first time this (tensor, version) is
    seen -- through one device-to-host copy, which also validates them."""
    hit = _KINDS_HOST.get(id(t))
    if hit is not None and hit[0]() is t and hit[1] == t._version:
        return hit[2]
    tup = tuple(int(k) for k in t.detach().cpu().tolist())
    _check_kinds(tup)
    _kinds_remember(t, tup)
    return tup


def kinds_tensor(kinds, device):
    """int32 device tensor of primitive kinds; rejects cones (sampling.py:39-45 is `pass`).  A tensor already on the
    device is validated once per (tensor object, version) -- one host copy the first time -- so that an unknown kind never
    reaches a kernel; a host list maps to one cached tensor per (list, device)."""
    device = torch.device(device)
    if isinstance(kinds, torch.Tensor):
        if kinds.device.type == device.type and kinds.dtype == torch.int32 and kinds.is_contiguous():
            kinds_host(kinds)
            return kinds
        kinds = kinds.detach().cpu().tolist()
    tup = tuple(int(k) for k in kinds)
    _check_kinds(tup)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    key = (tup, str(device))
    t = _KINDS_BY_TUPLE.get(key)
    if t is None:
        t = torch.tensor(tup, dtype=torch.int32, device=device)
        _KINDS_BY_TUPLE[key] = t
        _kinds_remember(t, tup)
    return t


def _seed_args(seed):
    """(host seed, device seed pointer) of the sampler entry points: `seed` is an int, or a device int64 tensor of
    one element (a step counter bumped on the stream: read by the kernel, so HIP-graph replays draw fresh points)."""
    if isinstance(seed, torch.Tensor):
        if not (seed.is_cuda and seed.dtype == torch.int64 and seed.numel() == 1):
            raise ValueError('a device seed must be a CUDA int64 tensor with one element')
        return 0, _lib.ptr(seed)
    return int(seed), None


class SampleFunction(Function):
    """Sampling.{sphere,cuboid}_sampling + transform_points + torch.cat over K primitives
    (sampling.py:11-37, train.py:105-120) -> points [B, K*n, 3]."""

    @staticmethod
    def forward(ctx, params, kinds, u, seed, sample_base, n):
        params = _f32c(params)
        B, K, S = params.shape
        kinds = kinds_tensor(kinds, params.device)
        assert S == PARAM_STRIDE and kinds.numel() == K
        if u is not None:
            u = _f32c(u)
            assert u.shape == (B, K, n, 3)
        points = torch.empty((B, K * n, 3), dtype=torch.float32, device=params.device)
        _lib.call('vpn_sample_fwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(u), int(seed), None, int(sample_base),
                                    B, K, n, _lib.ptr(points), _lib.stream())
        ctx.save_for_backward(params, kinds, u if u is not None else torch.empty(0, device=params.device))
        ctx.has_u = u is not None
        ctx.meta = (int(seed), int(sample_base), B, K, n)
        return points

    @staticmethod
    def backward(ctx, grad_points):
        params, kinds, u = ctx.saved_tensors
        seed, base, B, K, n = ctx.meta
        grad_points = _f32c(grad_points)
        grad_params = torch.empty_like(params)
        _lib.call('vpn_sample_bwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(u) if ctx.has_u else None, seed, None,
                                    base, B, K, n, _lib.ptr(grad_points), _lib.ptr(grad_params), _lib.stream())
        return grad_params, None, None, None, None, None


class TransformFunction(Function):
    """transform_points / rotate_points (transform.py:6-9, rotate.py:7-25): R(q) p (+ t)."""

    @staticmethod
    def forward(ctx, points, q, t):
        points, q = _f32c(points), _f32c(q)
        t = _f32c(t) if t is not None else None
        B, N, _ = points.shape
        out = torch.empty_like(points)
        _lib.call('vpn_transform_fwd', _lib.ptr(points), _lib.ptr(q), _lib.ptr(t), B, N, _lib.ptr(out),
                                       _lib.stream())
        ctx.save_for_backward(points, q)
        ctx.has_t = t is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        points, q = ctx.saved_tensors
        B, N, _ = points.shape
        grad_out = _f32c(grad_out)
        need_p, need_q, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_t and ctx.needs_input_grad[2]
        gp = torch.empty_like(points) if need_p else None
        gq = torch.empty_like(q) if need_q else None
        gt = torch.empty((B, 3), dtype=torch.float32, device=points.device) if need_t else None
        _lib.call('vpn_transform_bwd', _lib.ptr(points), _lib.ptr(q), _lib.ptr(grad_out), B, N, _lib.ptr(gp),
                                       _lib.ptr(gq), _lib.ptr(gt), _lib.stream())
        return gp, gq, gt


class MeshFunction(Function):
    """Vertices of all K primitives of all B samples in one launch (modules/meshing of the reference):
    params [B,K,10], templates [P,3] -> verts [B, sum_k P_kind(k), 3] in primitive order."""

    @staticmethod
    def forward(ctx, params, kinds, offsets, tpl_sphere, tpl_cuboid, ptot):
        params = _f32c(params)
        B, K, _ = params.shape
        kinds = kinds_tensor(kinds, params.device)
        ts = _f32c(tpl_sphere) if tpl_sphere is not None else None
        tc = _f32c(tpl_cuboid) if tpl_cuboid is not None else None
        verts = torch.empty((B, int(ptot), 3), dtype=torch.float32, device=params.device)
        _lib.call('vpn_mesh_fwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(offsets), _lib.ptr(ts), _lib.ptr(tc), B, K,
                  int(ptot), _lib.ptr(verts), _lib.stream())
        ctx.save_for_backward(params, kinds, offsets, *(x for x in (ts, tc) if x is not None))
        ctx.which = (ts is not None, tc is not None, int(ptot))
        return verts

    @staticmethod
    def backward(ctx, grad_verts):
        params, kinds, offsets, *tpls = ctx.saved_tensors
        has_s, has_c, ptot = ctx.which
        ts = tpls[0] if has_s else None
        tc = tpls[-1] if has_c else None
        B, K, _ = params.shape
        g = _f32c(grad_verts)
        gp = torch.empty_like(params)
        _lib.call('vpn_mesh_bwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(offsets), _lib.ptr(ts), _lib.ptr(tc), B, K, ptot,
                  _lib.ptr(g), _lib.ptr(gp), _lib.stream())
        return gp, None, None, None, None, None


_FACES = {}
_FACES_FP = {}


def faces_fingerprint(faces):
    """Content key of a face tensor: (shape, hash of its bytes), computed once per (tensor object, version).  Free for a
    host tensor; one device-to-host copy for a device tensor that is seen for the first time (TriangleMesh.to carries the
    key of the host tensor over, so meshes loaded from OBJ files never pay it)."""
    hit = _FACES_FP.get(id(faces))
    if hit is not None and hit[0]() is faces and hit[1] == faces._version:
        return hit[2]
    host = faces.detach().cpu().contiguous()
    fp = (tuple(host.shape), str(host.dtype), hash(host.numpy().tobytes()))
    faces_remember(faces, fp)
    return fp


def faces_remember(faces, fp):
    key = id(faces)
    _FACES_FP[key] = (weakref.ref(faces, lambda _r, key=key: _FACES_FP.pop(key, None)), faces._version, fp)


def faces_i32(faces, device):
    """[F,3] int32 contiguous device copy of a face tensor (the reference's are int64: meshing.py:38-39), one per CONTENT
    and device: keyed by the fingerprint, never by an address the allocator may hand to another tensor."""
    if faces.dtype == torch.int32 and faces.is_cuda and faces.is_contiguous():
        return faces
    key = (faces_fingerprint(faces), str(torch.device(device)))
    hit = _FACES.get(key)
    if hit is None:
        if len(_FACES) > 256:
            _FACES.clear()
        hit = faces.detach().to(device=device, dtype=torch.int32).contiguous()
        _FACES[key] = hit
    return hit


class MeshRasterFunction(Function):
    """Soft silhouette of triangle meshes WITHOUT primitives (the mesh input of vertex_renderer.py:20-24 as
    train_sphere.py:128 passes it): verts [B,P,3], faces [F,3] int32, cam [B,3] -> alpha [B,H,W]."""

    @staticmethod
    def forward(ctx, verts, faces, cam, H, W, sigma):
        verts, cam = _f32c(verts), _f32c(cam)
        B, P, _ = verts.shape
        F = faces.shape[0]
        assert faces.dtype == torch.int32 and faces.is_cuda and faces.is_contiguous() and cam.shape == (B, 3)
        dev = verts.device
        ws = torch.empty((_lib.lib().vpn_mesh_raster_workspace(B, P) // 4,), dtype=torch.float32, device=dev)
        alpha = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        _lib.call('vpn_mesh_raster_fwd', _lib.ptr(verts), _lib.ptr(faces), _lib.ptr(cam), B, P, F, H, W, float(sigma),
                  _lib.ptr(ws), _lib.ptr(alpha), _lib.stream())
        ctx.save_for_backward(verts, faces, cam, ws, alpha)
        ctx.meta = (B, P, F, H, W, float(sigma))
        return alpha

    @staticmethod
    def backward(ctx, grad_alpha):
        verts, faces, cam, ws, alpha = ctx.saved_tensors
        B, P, F, H, W, sigma = ctx.meta
        g = _f32c(grad_alpha)
        gv = torch.empty_like(verts)
        _lib.call('vpn_mesh_raster_bwd', _lib.ptr(verts), _lib.ptr(faces), _lib.ptr(cam), B, P, F, H, W, sigma, _lib.ptr(ws),
                  _lib.ptr(alpha), _lib.ptr(g), _lib.ptr(gv), _lib.stream())
        return gv, None, None, None, None, None


class MeshSampleFunction(Function):
    """kaolin's TriangleMesh.sample as train_sphere.py:76 uses it: n area-weighted uniform surface points per mesh.
    verts [B,P,3], faces [F,3] int32 -> points [B,n,3] (differentiable w.r.t. verts), face index [B,n] int32."""

    @staticmethod
    def forward(ctx, verts, faces, n, u, seed, mesh_base):
        verts = _f32c(verts)
        B, P, _ = verts.shape
        F = faces.shape[0]
        assert faces.dtype == torch.int32 and faces.is_cuda and faces.is_contiguous()
        dev = verts.device
        if u is not None:
            u = _f32c(u)
            assert u.shape == (B, n, 3)
        cdf = torch.empty((B, F), dtype=torch.float32, device=dev)
        points = torch.empty((B, n, 3), dtype=torch.float32, device=dev)
        fidx = torch.empty((B, n), dtype=torch.int32, device=dev)
        bary = torch.empty((B, n, 3), dtype=torch.float32, device=dev)
        _lib.call('vpn_mesh_sample_fwd', _lib.ptr(verts), _lib.ptr(faces), _lib.ptr(u), int(seed), int(mesh_base), B, P, F, int(n),
                  _lib.ptr(cdf), _lib.ptr(points), _lib.ptr(fidx), _lib.ptr(bary), _lib.stream())
        ctx.save_for_backward(faces, fidx, bary)
        ctx.meta = (B, P, F, int(n))
        ctx.mark_non_differentiable(fidx)
        return points, fidx

    @staticmethod
    def backward(ctx, grad_points, _grad_idx):
        faces, fidx, bary = ctx.saved_tensors
        B, P, F, n = ctx.meta
        g = _f32c(grad_points)
        gv = torch.empty((B, P, 3), dtype=torch.float32, device=g.device)
        _lib.call('vpn_mesh_sample_bwd', _lib.ptr(faces), _lib.ptr(fidx), _lib.ptr(bary), _lib.ptr(g), B, P, F, n, _lib.ptr(gv),
                  _lib.stream())
        return gv, None, None, None, None, None


class HeadPackFunction(Function):
    """restrict_range + split + restrict_volumes of the reference's model (vpnet_one_resnet.py:34-41, :67-85) fused:
    raw head outputs volumes [B,3K], rotates [B,4K], translates [B,3K] -> packed params [B,K,10]."""

    @staticmethod
    def forward(ctx, volumes, rotates, translates, is_sigmoid, clamp_min, clamp_max, restrict):
        volumes, rotates, translates = _f32c(volumes), _f32c(rotates), _f32c(translates)
        B = volumes.shape[0]
        assert volumes.shape[1] % 3 == 0
        K = volumes.shape[1] // 3
        assert rotates.shape == (B, 4 * K) and translates.shape == (B, 3 * K)
        r = [float(x) for x in restrict]
        assert len(r) == 3
        params = torch.empty((B, K, PARAM_STRIDE), dtype=torch.float32, device=volumes.device)
        _lib.call('vpn_head_pack_fwd', _lib.ptr(volumes), _lib.ptr(rotates), _lib.ptr(translates), B, K,
                  int(bool(is_sigmoid)), float(clamp_min), float(clamp_max), r[0], r[1], r[2], _lib.ptr(params),
                  _lib.stream())
        ctx.save_for_backward(volumes, rotates, translates)
        ctx.cfg = (B, K, int(bool(is_sigmoid)), float(clamp_min), float(clamp_max), r)
        return params

    @staticmethod
    def backward(ctx, grad_params):
        volumes, rotates, translates = ctx.saved_tensors
        B, K, sig, cmin, cmax, r = ctx.cfg
        g = _f32c(grad_params)
        gv = torch.empty_like(volumes) if ctx.needs_input_grad[0] else None
        gq = torch.empty_like(rotates) if ctx.needs_input_grad[1] else None
        gt = torch.empty_like(translates) if ctx.needs_input_grad[2] else None
        _lib.call('vpn_head_pack_bwd', _lib.ptr(volumes), _lib.ptr(rotates), _lib.ptr(translates), _lib.ptr(g), B, K, sig,
                  cmin, cmax, r[0], r[1], r[2], _lib.ptr(gv), _lib.ptr(gq), _lib.ptr(gt), _lib.stream())
        return gv, gq, gt, None, None, None, None


class CameraTransformFunction(Function):
    """view_to_obj_points / obj_to_view_points (modules/transform/transform.py:21-73) in one launch.
    dists, elevs, azims, angles are dataset values (dataset.py:145-165): constants for autograd."""

    @staticmethod
    def forward(ctx, points, dists, elevs, azims, angles, to_object):
        points = _f32c(points)
        B, N, _ = points.shape
        cam = [_f32c(c.reshape(-1)) for c in (dists, elevs, azims)]
        ang = _f32c(angles.reshape(-1)) if angles is not None else None
        for c in cam + ([ang] if ang is not None else []):
            if c.numel() != B:
                raise ValueError('camera arguments must hold one value per sample')
        out = torch.empty_like(points)
        _lib.call('vpn_camera_transform_fwd', _lib.ptr(points), _lib.ptr(cam[0]), _lib.ptr(cam[1]), _lib.ptr(cam[2]),
                  _lib.ptr(ang), B, N, int(bool(to_object)), _lib.ptr(out), _lib.stream())
        ctx.save_for_backward(*cam, *([ang] if ang is not None else []))
        ctx.to_object = int(bool(to_object))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        saved = ctx.saved_tensors
        d, e, a = saved[:3]
        ang = saved[3] if len(saved) > 3 else None
        g = _f32c(grad_out)
        B, N, _ = g.shape
        gp = torch.empty_like(g)
        _lib.call('vpn_camera_transform_bwd', _lib.ptr(g), _lib.ptr(d), _lib.ptr(e), _lib.ptr(a), _lib.ptr(ang), B, N,
                  ctx.to_object, _lib.ptr(gp), _lib.stream())
        return gp, None, None, None, None, None


class ChamferFunction(Function):
    """ChamferDistanceLoss.forward up to the per-sample loss (chamfer_distance.py:14-28).
    Returns loss_b [B]; the caller takes .mean() unless each_batch (chamfer_distance.py:30)."""

    @staticmethod
    def forward(ctx, p1, p2, w1, w2):
        p1, p2 = _f32c(p1), _f32c(p2)
        B, N, _ = p1.shape
        M = p2.shape[1]
        dev = p1.device
        d1 = torch.empty((B, N), dtype=torch.float32, device=dev)
        d2 = torch.empty((B, M), dtype=torch.float32, device=dev)
        i1 = torch.empty((B, N), dtype=torch.int32, device=dev)
        i2 = torch.empty((B, M), dtype=torch.int32, device=dev)
        loss_b = torch.empty((B,), dtype=torch.float32, device=dev)
        s = _lib.stream()
        ws = torch.empty((_lib.lib().vpn_chamfer_workspace(B, N, M) // 4,), dtype=torch.float32, device=dev)
        _lib.call('vpn_chamfer_fwd_ws', _lib.ptr(p1), _lib.ptr(p2), B, N, M, _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2),
                  _lib.ptr(i2), _lib.ptr(ws), ws.numel() * 4, 0, s)
        _lib.call('vpn_chamfer_loss', _lib.ptr(d1), _lib.ptr(d2), B, N, M, float(w1), float(w2), _lib.ptr(loss_b), s)
        ctx.save_for_backward(p1, p2, d1, i1, d2, i2)
        ctx.w = (float(w1), float(w2))
        return loss_b

    @staticmethod
    def backward(ctx, grad_loss_b):
        p1, p2, d1, i1, d2, i2 = ctx.saved_tensors
        B, N, _ = p1.shape
        M = p2.shape[1]
        g = _f32c(grad_loss_b)
        g1 = torch.empty_like(p1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(p2) if ctx.needs_input_grad[1] else None
        _lib.call('vpn_chamfer_bwd', _lib.ptr(p1), _lib.ptr(p2), _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2),
                                     _lib.ptr(i2), _lib.ptr(g), B, N, M, ctx.w[0], ctx.w[1], _lib.ptr(g1),
                                     _lib.ptr(g2), _lib.stream())
        return g1, g2, None, None


CHAMFER_MODES = {'auto': 0, 'brute': 1, 'pruned': 2, 'mfma': 3, 'mfma32': 4, 'sorted': 5, 'mfma16': 6}


def chamfer_nn(p1, p2, mode='auto'):
    """Nearest-neighbour distances and indices in both directions (no autograd):
    (dist1 [B,N], idx1 [B,N] int32, dist2 [B,M], idx2 [B,M] int32).  mode: 'auto' | 'brute' | 'pruned'
    (same results bit for bit; 'pruned' Morton-sorts the clouds and skips far target chunks, 'mfma' / 'mfma32'
    filter on the bf16 / fp32 matrix instructions and finish exactly)."""
    p1, p2 = _f32c(p1.detach()), _f32c(p2.detach())
    B, N, _ = p1.shape
    M = p2.shape[1]
    dev = p1.device
    d1 = torch.empty((B, N), dtype=torch.float32, device=dev)
    d2 = torch.empty((B, M), dtype=torch.float32, device=dev)
    i1 = torch.empty((B, N), dtype=torch.int32, device=dev)
    i2 = torch.empty((B, M), dtype=torch.int32, device=dev)
    ws = torch.empty((_lib.lib().vpn_chamfer_workspace(B, N, M) // 4,), dtype=torch.float32, device=dev)
    _lib.call('vpn_chamfer_fwd_ws', _lib.ptr(p1), _lib.ptr(p2), B, N, M, _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2),
              _lib.ptr(i2), _lib.ptr(ws), ws.numel() * 4, CHAMFER_MODES[mode], _lib.stream())
    return d1, i1, d2, i2


class EmdFunction(Function):
    """emdFunction (modules/loss/emd/emd_module.py:29-70) on vpn_emd_fwd / vpn_emd_bwd: auction
    approximation of the Earth Mover's Distance.  Returns (dist [B,n] squared distance to the assigned
    point, assignment [B,n] int32).  The reference's nine scratch tensors are one workspace here."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, eps, iters, max_group=None):
        """max_group: workgroups per sample (None: VPN_EMD_GROUP or automatic; 1: no inter-workgroup barrier -- forced
        when other streams share the GPU (VPN_CONCURRENT=1), because the group barrier needs the whole grid resident)."""
        B, n, _ = xyz1.size()
        assert n == xyz2.size(1)                       # emd_module.py:36-37
        assert B == xyz2.size(0)
        xyz1, xyz2 = _f32c(xyz1), _f32c(xyz2)
        dev = xyz1.device
        dist = torch.empty((B, n), dtype=torch.float32, device=dev)
        assignment = torch.empty((B, n), dtype=torch.int32, device=dev)
        ws = torch.empty((max(1, _lib.lib().vpn_emd_workspace(B, n) // 4),), dtype=torch.float32, device=dev)
        if max_group is None:
            max_group = 1 if CONCURRENT_BRANCHES else int(os.environ.get('VPN_EMD_GROUP', '0'))
        _lib.call('vpn_emd_fwd', _lib.ptr(xyz1), _lib.ptr(xyz2), B, n, float(eps), int(iters), _lib.ptr(dist),
                  _lib.ptr(assignment), _lib.ptr(ws), int(max_group), _lib.stream())
        ctx.save_for_backward(xyz1, xyz2, assignment)
        ctx.mark_non_differentiable(assignment)
        return dist, assignment

    @staticmethod
    def backward(ctx, graddist, gradidx):
        xyz1, xyz2, assignment = ctx.saved_tensors
        B, n, _ = xyz1.shape
        g = _f32c(graddist)
        g1 = torch.empty_like(xyz1)
        _lib.call('vpn_emd_bwd', _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(g), _lib.ptr(assignment), B, n,
                  _lib.ptr(g1), _lib.stream())
        g2 = torch.zeros_like(xyz2) if ctx.needs_input_grad[1] else None     # emd_module.py:67
        return g1, g2, None, None, None


class RasterFunction(Function):
    """Primitive soft raster behind VertexRenderer.render (vertex_renderer.py:14-26):
    params [B,K,10], cam [B,3] = (dist, elev_deg, azim_deg) -> alpha, depth [B,H,W]."""

    @staticmethod
    def forward(ctx, params, kinds, cam, H, W, sigma, gamma, z_far):
        params, cam = _f32c(params), _f32c(cam)
        B, K, S = params.shape
        kinds = kinds_tensor(kinds, params.device)
        assert S == PARAM_STRIDE and kinds.numel() == K and cam.shape == (B, 3)
        dev = params.device
        alpha = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        aux = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
        rec = torch.empty((_lib.lib().vpn_raster_records_size(B, K, H, W) // 4,), dtype=torch.float32, device=dev)
        _lib.call('vpn_raster_fwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W,
                  float(sigma), float(gamma), float(z_far), _lib.ptr(alpha), _lib.ptr(depth), _lib.ptr(aux),
                  _lib.ptr(rec), _lib.stream())
        ctx.save_for_backward(params, kinds, cam, aux, rec)
        ctx.meta = (B, K, H, W, float(sigma), float(gamma), float(z_far))
        return alpha, depth

    @staticmethod
    def backward(ctx, grad_alpha, grad_depth):
        params, kinds, cam, aux, rec = ctx.saved_tensors
        B, K, H, W, sigma, gamma, z_far = ctx.meta
        ga = _f32c(grad_alpha) if grad_alpha is not None else None
        gd = _f32c(grad_depth) if grad_depth is not None else None
        ws = torch.empty((_lib.lib().vpn_raster_bwd_workspace(B, K, H, W) // 4,), dtype=torch.float32, device=params.device)
        grad_params = torch.empty_like(params)
        _lib.call('vpn_raster_bwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, sigma, gamma,
                  z_far, _lib.ptr(aux), _lib.ptr(rec), _lib.ptr(ga), _lib.ptr(gd), _lib.ptr(ws),
                  _lib.ptr(grad_params), _lib.stream())
        return grad_params, None, None, None, None, None, None, None


class RasterLossFunction(Function):
    """SilhouetteLoss.forward (silhouette.py:13-23) fused into the raster: render + L1/MSE mean against
    the GT silhouette (+ optional L1 depth loss) without materialising the images.
    Returns a [2] tensor: (silhouette loss, depth loss)."""

    @staticmethod
    def forward(ctx, params, kinds, cam, gt_sil, gt_depth, H, W, sigma, gamma, z_far, sil_mse):
        params, cam = _f32c(params), _f32c(cam)
        B, K, S = params.shape
        kinds = kinds_tensor(kinds, params.device)
        assert S == PARAM_STRIDE and kinds.numel() == K and cam.shape == (B, 3)
        if gt_sil is not None:
            gt_sil = _f32c(gt_sil).reshape(B, H, W)
        if gt_depth is not None:
            gt_depth = _f32c(gt_depth).reshape(B, H, W)
        dev = params.device
        L = _lib.lib()
        aux = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
        rec = torch.empty((L.vpn_raster_records_size(B, K, H, W) // 4,), dtype=torch.float32, device=dev)
        lws = torch.empty((L.vpn_raster_loss_workspace(B, H, W) // 4,), dtype=torch.float32, device=dev)
        losses = torch.empty((4,), dtype=torch.float32, device=dev)
        _lib.call('vpn_raster_loss_fwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, float(sigma),
                  float(gamma), float(z_far), _lib.ptr(gt_sil), _lib.ptr(gt_depth), int(bool(sil_mse)), _lib.ptr(aux),
                  _lib.ptr(rec), _lib.ptr(lws), _lib.ptr(losses), _lib.stream())
        losses = losses[:2]
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(params, kinds, cam, aux, rec, gt_sil if gt_sil is not None else empty,
                              gt_depth if gt_depth is not None else empty)
        ctx.meta = (B, K, H, W, float(sigma), float(gamma), float(z_far), int(bool(sil_mse)),
                    gt_sil is not None, gt_depth is not None)
        return losses

    @staticmethod
    def backward(ctx, grad_losses):
        params, kinds, cam, aux, rec, gt_sil, gt_depth = ctx.saved_tensors
        B, K, H, W, sigma, gamma, z_far, sil_mse, has_sil, has_depth = ctx.meta
        g = _f32c(grad_losses)
        ws = torch.empty((_lib.lib().vpn_raster_bwd_workspace(B, K, H, W) // 4,), dtype=torch.float32,
                         device=params.device)
        grad_params = torch.empty_like(params)
        _lib.call('vpn_raster_loss_bwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, sigma, gamma,
                  z_far, _lib.ptr(aux), _lib.ptr(rec), _lib.ptr(gt_sil) if has_sil else None,
                  _lib.ptr(gt_depth) if has_depth else None, sil_mse, _lib.ptr(g), _lib.ptr(ws),
                  _lib.ptr(grad_params), 0, _lib.stream())
        return (grad_params,) + (None,) * 10


class RasterTotalFunction(Function):
    """total_img = w_sil * SilhouetteLoss + w_dep * L1(depth), forward and backward in one pass over the image
    (vpn_raster_total_fwd: no aux tensor, GT read once, the gradient partials are produced by the forward launch;
    backward is the small finishing kernel).  Returns (silhouette loss, depth loss, total_img); only the total is
    differentiable."""

    @staticmethod
    def forward(ctx, params, kinds, cam, gt_sil, gt_depth, H, W, sigma, gamma, z_far, sil_mse, w_sil, w_dep):
        params, cam = _f32c(params), _f32c(cam)
        B, K, S = params.shape
        kinds = kinds_tensor(kinds, params.device)
        assert S == PARAM_STRIDE and kinds.numel() == K and cam.shape == (B, 3)
        gt_sil = _f32c(gt_sil).reshape(B, H, W) if gt_sil is not None else None
        gt_depth = _f32c(gt_depth).reshape(B, H, W) if gt_depth is not None else None
        dev = params.device
        L = _lib.lib()
        s = _lib.stream()
        rec = torch.empty((L.vpn_raster_records_size(B, K, H, W) // 4,), dtype=torch.float32, device=dev)
        lws = torch.empty((L.vpn_raster_loss_workspace(B, H, W) // 4,), dtype=torch.float32, device=dev)
        ws = torch.empty((L.vpn_raster_bwd_workspace(B, K, H, W) // 4,), dtype=torch.float32, device=dev)
        losses = torch.empty((4,), dtype=torch.float32, device=dev)
        # render, image losses, gradient partials and the loss finalisation: one launch after the record launch
        _lib.call('vpn_raster_total_fwd_fin', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, float(sigma),
                  float(gamma), float(z_far), _lib.ptr(gt_sil), _lib.ptr(gt_depth), int(bool(sil_mse)), float(w_sil),
                  float(w_dep), _lib.ptr(rec), _lib.ptr(lws), _lib.ptr(ws), 0, None, 0, 0, 0, 0.0, 0.0, 0.0,
                  _lib.ptr(losses), None, None, None, s)
        ctx.save_for_backward(params, cam, rec, ws)
        ctx.meta = (B, K, H, W)
        sil, dep, tot, _ = losses.unbind(0)
        ctx.mark_non_differentiable(sil, dep)
        ctx.set_materialize_grads(False)       # no zero-filled gradients for the two reported values
        return sil, dep, tot

    @staticmethod
    def backward(ctx, _g_sil, _g_dep, grad_total):
        params, cam, rec, ws = ctx.saved_tensors
        B, K, H, W = ctx.meta
        if grad_total is None:
            return (None,) * 13
        g = _f32c(grad_total).reshape(1)
        grad_params = torch.empty_like(params)
        _lib.call('vpn_raster_total_bwd', _lib.ptr(params), _lib.ptr(cam), B, K, H, W, _lib.ptr(rec), _lib.ptr(ws),
                  _lib.ptr(g), _lib.ptr(grad_params), 0, _lib.stream())
        return (grad_params,) + (None,) * 12


TILE_ORDER = os.environ.get('VPN_TILE_ORDER', '1') != '0'     # 0: position-based launch order of the tile waves (A/B switch)
_PATTERNS = {}
FUSED_BWD_MAX_GT = 7680       # vpn_sample_chamfer_bwd keeps per-wave match lists of the GT points in LDS (include/vpn_hip.h)
_SIDE = {}
# optionally run the raster branch of HotPathLossFunction on a second HIP stream (VPN_CONCURRENT=1)
# TrainStepLossFunction: the auction on a second stream, beside the scans and the raster (measured at C5: 1.10 -> 0.94 ms per
# step; VPN_EMD_SIDE=0 puts it back in line)
EMD_SIDE_STREAM = os.environ.get('VPN_EMD_SIDE', '1') == '1'
CONCURRENT_BRANCHES = os.environ.get('VPN_CONCURRENT', '0') == '1'   # measured: no gain at C3 (each kernel already fills the GPU)


def _side_stream(dev):
    key = str(dev)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def _grad_pattern(B, w_cd, dev):
    """d total / d loss_b[0..B): constant per configuration, cached on the device (created outside any graph capture
    by the first eager call)."""
    key = (B, float(w_cd), str(dev))
    if key not in _PATTERNS:
        _PATTERNS[key] = torch.full((B,), w_cd / B, dtype=torch.float32, device=dev)
    return _PATTERNS[key]


class HotPathLossFunction(Function):
    """One training-step loss of the reference's hot path in a single autograd node (train.py:243-262):
        total = w_cd * ChamferDistanceLoss(sample(params), gt_points; cd_w1, cd_w2) + w_sil * SilhouetteLoss
                + w_depth * L1(depth)
    Forward: sampler (+ raster records) -> Chamfer scans -> raster forward+backward pass (image losses and their gradient
    partials in one launch) -> loss finalisation (per-sample Chamfer losses, image losses, total).  Backward:
    Chamfer + sampler backward (writes d/dparams) -> raster finishing kernel (adds to it).  No intermediate ever goes
    through an ATen kernel.  Returns three scalars (silhouette loss, depth loss, total); only the total is
    differentiable (the first two are reported values: they are marked non-differentiable, so `out[0].backward()`
    raises instead of returning a wrong gradient).
    `seed`: int, or a device int64 tensor of one element read by the kernels (see _seed_args).
    cd_w1 / cd_w2 = config.CD_W1 / CD_W2 (chamfer_distance.py:10), sil_mse = SILHOUETTE_LOSS_FUNC != 'L1'."""

    @staticmethod
    def forward(ctx, params, kinds, cam, gt_points, gt_sil, gt_depth, n, seed, sample_base, H, W, sigma, gamma,
                z_far, w_cd, w_sil, w_depth, cd_w1=1.0, cd_w2=1.0, sil_mse=False, advance_seed=False):
        params, cam, gt_points = _f32c(params), _f32c(cam), _f32c(gt_points)
        B, K, _ = params.shape
        M = gt_points.shape[1]
        N = K * n
        dev = params.device
        kinds = kinds_tensor(kinds, dev)
        assert kinds.numel() == K and cam.shape == (B, 3) and gt_points.shape[0] == B
        L = _lib.lib()
        s = _lib.stream()
        seed_host, seed_dev = _seed_args(seed)
        cd_w1, cd_w2, sil_mse = float(cd_w1), float(cd_w2), int(bool(sil_mse))
        gt_sil = _f32c(gt_sil).reshape(B, H, W) if gt_sil is not None else None
        gt_depth = _f32c(gt_depth).reshape(B, H, W) if gt_depth is not None else None
        rec = torch.empty((L.vpn_raster_records_size(B, K, H, W) // 4,), dtype=torch.float32, device=dev)
        lws = torch.empty((L.vpn_raster_loss_workspace(B, H, W) // 4,), dtype=torch.float32, device=dev)
        rws = torch.empty((L.vpn_raster_bwd_workspace(B, K, H, W) // 4,), dtype=torch.float32, device=dev)
        losses = torch.empty((4,), dtype=torch.float32, device=dev)

        def raster_branch(stream, records_ready):
            _lib.call('vpn_raster_total_fwd', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, float(sigma),
                      float(gamma), float(z_far), _lib.ptr(gt_sil), _lib.ptr(gt_depth), sil_mse, float(w_sil),
                      float(w_depth), _lib.ptr(rec), _lib.ptr(lws), _lib.ptr(rws), records_ready, stream)

        if advance_seed and seed_dev is None:
            raise ValueError('advance_seed needs a device seed (a CUDA int64 tensor of one element)')

        main = torch.cuda.current_stream()
        # optionally the raster branch (independent of the sampler + Chamfer branch until the finalisation) runs on a
        # side stream; fork / join is captured into HIP graphs as such
        side = _side_stream(dev) if CONCURRENT_BRANCHES else None
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                raster_branch(_lib.stream(), 0)
                for t in (params, kinds, cam, gt_sil, gt_depth, rec, lws, rws):
                    if t is not None:
                        t.record_stream(side)
        points = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        cws = torch.empty((L.vpn_chamfer_workspace(B, N, M) // 4,), dtype=torch.float32, device=dev)
        chamfer_mode = 0
        ntile = ((W + 15) // 16) * ((H + 15) // 16)
        use_order = False
        if side is None:        # one stream: the sampler's launch also writes the raster records of the same primitives
            # ... and, when the Chamfer scan will be the matrix-pipe filter, that filter's features of both clouds
            fused = bool(L.vpn_hotpath_fused_features(B, K, n, M))
            # the scan launch can carry a rider that tests the tiles against the primitives and sorts them by weight
            use_order = fused and TILE_ORDER and K <= 64 and ntile <= 16384 and K * 84 + (K + 2) * 4 + ntile <= 24576
            _lib.call('vpn_hotpath_sample_fwd', _lib.ptr(params), _lib.ptr(kinds), None, seed_host, seed_dev,
                      int(sample_base), B, K, n, _lib.ptr(points), _lib.ptr(cam), H, W, float(sigma),
                      _lib.ptr(rec),
                      _lib.ptr(lws), _lib.ptr(gt_points), M, _lib.ptr(cws) if fused else None, cws.numel() * 4, s)
            chamfer_mode = 7 if fused else 0
        else:
            _lib.call('vpn_sample_fwd', _lib.ptr(params), _lib.ptr(kinds), None, seed_host, seed_dev, int(sample_base), B,
                      K, n, _lib.ptr(points), s)
        d1 = torch.empty((B, N), dtype=torch.float32, device=dev)
        d2 = torch.empty((B, M), dtype=torch.float32, device=dev)
        i1 = torch.empty((B, N), dtype=torch.int32, device=dev)
        i2 = torch.empty((B, M), dtype=torch.int32, device=dev)
        fused_fin = side is None and chamfer_mode == 7
        order = None
        if use_order:
            # the scan launch also prepares the raster's tiles (a rider in its tail): one entry per tile wave -- which tile
            # (heaviest first), which primitives it sees, which quadrants each of them reaches -- instead of every tile wave
            # finding that out for itself
            order = torch.empty((L.vpn_raster_order_size(B, H, W) // 8,), dtype=torch.int64, device=dev)   # 48-byte tile entries
            _lib.call('vpn_hotpath_chamfer_fwd', _lib.ptr(points), _lib.ptr(gt_points), B, N, M, _lib.ptr(d1), _lib.ptr(i1),
                      _lib.ptr(d2), _lib.ptr(i2), _lib.ptr(cws), cws.numel() * 4, chamfer_mode, _lib.ptr(rec), K, H, W,
                      _lib.ptr(order), s)
        else:
            _lib.call('vpn_chamfer_fwd_ws', _lib.ptr(points), _lib.ptr(gt_points), B, N, M, _lib.ptr(d1), _lib.ptr(i1),
                      _lib.ptr(d2), _lib.ptr(i2), _lib.ptr(cws), cws.numel() * 4, chamfer_mode, s)
        if fused_fin:
            # raster forward+backward pass and the loss finalisation (image losses, per-sample Chamfer terms from the
            # scan's per-workgroup sums, total, optional advance of the step counter) in ONE launch
            _lib.call('vpn_raster_total_fwd_fin', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, float(sigma),
                      float(gamma), float(z_far), _lib.ptr(gt_sil), _lib.ptr(gt_depth), sil_mse, float(w_sil),
                      float(w_depth), _lib.ptr(rec), _lib.ptr(lws), _lib.ptr(rws), 1, _lib.ptr(cws), cws.numel() * 4, N, M,
                      cd_w1, cd_w2, float(w_cd), _lib.ptr(losses), None, seed_dev if advance_seed else None, _lib.ptr(order), s)
        else:
            if side is not None:
                main.wait_stream(side)
            else:
                raster_branch(s, 1)
            _lib.call('vpn_loss_finalize', _lib.ptr(lws), B, H, W, _lib.ptr(d1), _lib.ptr(d2), N, M, cd_w1, cd_w2, float(w_cd),
                      float(w_sil), float(w_depth), _lib.ptr(losses), None, s)
        pattern = _grad_pattern(B, w_cd, dev)
        empty = torch.empty(0, device=dev)
        seed_t = seed if isinstance(seed, torch.Tensor) else empty
        if advance_seed and not fused_fin:
            # the side-stream / unfused form advances the caller's counter here; when the sampler launch did not keep the
            # seed it used (side stream), the backward must not read the advanced counter: it gets a copy of the old one
            if not (side is None and seed_dev is not None):
                seed_t = seed.clone()
            seed.add_(1)
        # the sampler's launch of the one-stream path keeps the seed it used at loss_ws + 8: backward reads it from there,
        # whatever has happened to the caller's counter since
        seed_saved = side is None and seed_dev is not None
        ctx.save_for_backward(params, kinds, cam, gt_points, points, d1, i1, d2, i2, rec, rws, pattern,
                              lws if seed_saved else seed_t)
        ctx.meta = (B, K, n, M, H, W, 0 if seed_saved else seed_host, seed_dev is not None, int(sample_base), cd_w1, cd_w2,
                    float(w_cd) / B, seed_saved)
        sil, dep, tot, _ = losses.unbind(0)
        ctx.mark_non_differentiable(sil, dep)
        ctx.set_materialize_grads(False)       # no zero-filled gradients for the two reported (non-differentiable) values
        return sil, dep, tot

    @staticmethod
    def backward(ctx, _g_sil, _g_dep, grad_total):
        (params, kinds, cam, gt_points, points, d1, i1, d2, i2, rec, rws, pattern, seed_t) = ctx.saved_tensors
        B, K, n, M, H, W, seed, has_seed_dev, base, cd_w1, cd_w2, w_cd_over_b, seed_saved = ctx.meta
        N = K * n
        s = _lib.stream()
        seed_dev = None
        if seed_saved:                                          # seed_t is the loss workspace: effective seed at byte 8
            seed_dev = ctypes.c_void_p(seed_t.data_ptr() + 8)
        elif has_seed_dev:
            seed_dev = _lib.ptr(seed_t)
        if grad_total is None:                                  # the total was not used (set_materialize_grads(False))
            return (None,) * 21
        grad_total = _f32c(grad_total).reshape(1)
        # Chamfer backward and sampler backward in one launch: the [B,N,3] point gradient never exists
        grad_params = torch.empty_like(params)
        if M <= FUSED_BWD_MAX_GT:          # ... and the raster's finishing step rides in the same launch
            # d total / d loss_b = (w_cd / B) * grad_total for every sample: the constant goes into the two Chamfer
            # weights and the kernel reads grad_total itself (no ATen kernel between autograd and the launch)
            _lib.call('vpn_hotpath_bwd', _lib.ptr(params), _lib.ptr(kinds), None, seed, seed_dev, base, B, K, n,
                      _lib.ptr(points), _lib.ptr(gt_points), M, _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2), _lib.ptr(i2),
                      None, cd_w1 * w_cd_over_b, cd_w2 * w_cd_over_b, _lib.ptr(cam), H, W, _lib.ptr(rec), _lib.ptr(rws),
                      _lib.ptr(grad_total), _lib.ptr(grad_params), s)
            return (grad_params,) + (None,) * 20
        else:                                                   # GT clouds beyond the fused kernel's LDS match lists
            gvec = pattern * grad_total                         # d total / d loss_b [B]
            grad_points = torch.empty_like(points)
            _lib.call('vpn_chamfer_bwd', _lib.ptr(points), _lib.ptr(gt_points), _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2),
                      _lib.ptr(i2), _lib.ptr(gvec), B, N, M, cd_w1, cd_w2, _lib.ptr(grad_points), None, s)
            _lib.call('vpn_sample_bwd', _lib.ptr(params), _lib.ptr(kinds), None, seed, seed_dev, base, B, K, n,
                      _lib.ptr(grad_points), _lib.ptr(grad_params), s)
        # the raster's gradient partials were produced by the forward launch: chain rule x upstream gradient, added
        _lib.call('vpn_raster_total_bwd', _lib.ptr(params), _lib.ptr(cam), B, K, H, W, _lib.ptr(rec), _lib.ptr(rws),
                  _lib.ptr(grad_total), _lib.ptr(grad_params), 1, s)
        return (grad_params,) + (None,) * 20


_CAMS = {}


def _view_camera(B, dev):
    """cam [B,3] = (dist 1, elev 0, azim 0): the view-centred camera of train.py:172-174, cached per (B, device)."""
    key = (B, str(dev))
    if key not in _CAMS:
        _CAMS[key] = torch.tensor([[1.0, 0.0, 0.0]], dtype=torch.float32, device=dev).expand(B, 3).contiguous()
    return _CAMS[key]


class TrainStepLossFunction(Function):
    """The loss of one training iteration of the reference (train.py:243-262) as ONE autograd node, BASELINE config C5:

        total = L_VIEW_CD * ChamferDistanceLoss(pred, view_center_points)                          train.py:160
              + L_CAN_CD  * ChamferDistanceLoss(view_to_obj_points(pred, ...), canonical_points)   train.py:158-161
              + L_SIL     * SilhouetteLoss(primitives, silhouettes; dist 1, elev 0, azim 0)        train.py:169-176
              + L_VP_DIV  * VPDiverseLoss(translates, view_center_points)                          train.py:185
              + L_EMD     * sqrt(EarthMoverDistanceLoss(pred, view_center_points, eps, iters)[0]).mean()   train.py:193-195

    with pred = sample_predict_points (train.py:105-120) of params [B,K,10].  Forward: the hot path's three launches
    (sampler + raster records + Chamfer features, Chamfer scan + tile rider, raster with the loss finalisation), the
    auction, the object-centred cloud (camera transform, its Chamfer scan), the VP-diversity neighbours, one reduction;
    backward: ONE launch (vpn_trainstep_bwd).  No ATen kernel runs in either.  `weights` = (L_VIEW_CD, L_CAN_CD, L_SIL,
    L_VP_DIV, L_EMD) of config.py:13-17; a zero weight skips that term's backward (the reference multiplies it by 0), and
    L_SIL = 0 skips the render like train.py:167.  Returns six scalars (the five weighted terms, then the total); only the
    total is differentiable: backward through `out[5]` (the terms are marked non-differentiable)."""

    @staticmethod
    def forward(ctx, params, kinds, gt_view, gt_canon, gt_sil, dists, elevs, azims, angles, n, seed, sample_base, H, W,
                weights, eps=0.005, iters=50, advance_seed=False, cd_w1=1.0, cd_w2=1.0, sil_mse=False):
        from . import config
        params, gt_view = _f32c(params), _f32c(gt_view)
        B, K, _ = params.shape
        M = gt_view.shape[1]
        N = K * n
        dev = params.device
        kinds = kinds_tensor(kinds, dev)
        w_view, w_can, w_sil, w_div, w_emd = (float(x) for x in weights)
        cd_w1, cd_w2, sil_mse = float(cd_w1), float(cd_w2), int(bool(sil_mse))
        L = _lib.lib()
        s = _lib.stream()
        seed_host, seed_dev = _seed_args(seed)
        if advance_seed and seed_dev is None:
            raise ValueError('advance_seed needs a device seed (a CUDA int64 tensor of one element)')
        if w_emd and N != M:
            raise ValueError('the EMD term needs as many predicted as ground-truth points (emd_module.py:36): %d vs %d' % (N, M))
        if not L.vpn_hotpath_fused_features(B, K, n, M):
            raise ValueError('TrainStepLossFunction needs a shape the fused sampler / Chamfer path takes (K <= 64, large clouds)')
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        render = w_sil != 0.0 and gt_sil is not None
        cam = _view_camera(B, dev)
        gt_sil = _f32c(gt_sil).reshape(B, H, W) if render else None
        sigma, gamma, z_far = config.RASTER_SIGMA, config.RASTER_GAMMA, config.RASTER_Z_FAR
        # the render's buffers exist (tiny) even when the silhouette term is off: the sampler launch writes the records
        Hr, Wr = (H, W) if render else (16, 16)
        rec = torch.empty((L.vpn_raster_records_size(B, K, Hr, Wr) // 4,), **f32)
        lws = torch.empty((L.vpn_raster_loss_workspace(B, Hr, Wr) // 4,), **f32)
        rws = torch.empty((L.vpn_raster_bwd_workspace(B, K, Hr, Wr) // 4,), **f32)
        hot = torch.empty((4,), **f32)
        points = torch.empty((B, N, 3), **f32)
        cws = torch.empty((L.vpn_chamfer_workspace(B, N, M) // 4,), **f32)
        d1, d2 = torch.empty((B, N), **f32), torch.empty((B, M), **f32)
        i1, i2 = torch.empty((B, N), **i32), torch.empty((B, M), **i32)
        # ---- hot path: sampler (+ records + features) -> view-centred Chamfer (+ tile rider) -> raster + finalisation
        _lib.call('vpn_hotpath_sample_fwd', _lib.ptr(params), _lib.ptr(kinds), None, seed_host, seed_dev, int(sample_base), B, K, n,
                  _lib.ptr(points), _lib.ptr(cam), Hr, Wr, float(sigma), _lib.ptr(rec), _lib.ptr(lws), _lib.ptr(gt_view), M,
                  _lib.ptr(cws), cws.numel() * 4, s)
        # ---- EMD auction on the sampled cloud (train.py:193).  On a second stream when EMD_SIDE_STREAM: it needs only the
        #      sampler's points, nothing needs it before the final sums, and its workgroups (one per CU, 16 waves) leave
        #      every CU half of its wave slots and 32 KB of LDS -- the Chamfer scans and the raster run beside it
        emd_dist = emd_assign = None
        side = None
        if w_emd:
            emd_dist = torch.empty((B, N), **f32)
            emd_assign = torch.empty((B, N), **i32)
            ews = torch.empty((max(1, L.vpn_emd_workspace(B, N) // 4),), **f32)
            es = s
            if EMD_SIDE_STREAM:
                main = torch.cuda.current_stream()
                side = _side_stream(dev)
                side.wait_stream(main)
                es = ctypes.c_void_p(side.cuda_stream)
            _lib.call('vpn_emd_fwd', _lib.ptr(points), _lib.ptr(gt_view), B, N, float(eps), int(iters), _lib.ptr(emd_dist),
                      _lib.ptr(emd_assign), _lib.ptr(ews), 1 if CONCURRENT_BRANCHES else int(os.environ.get('VPN_EMD_GROUP', '0')), es)
        ntile = ((Wr + 15) // 16) * ((Hr + 15) // 16)
        use_order = render and TILE_ORDER and K <= 64 and ntile <= 16384 and K * 84 + (K + 2) * 4 + ntile <= 24576
        order = None
        if use_order:
            order = torch.empty((L.vpn_raster_order_size(B, Hr, Wr) // 8,), dtype=torch.int64, device=dev)
            _lib.call('vpn_hotpath_chamfer_fwd', _lib.ptr(points), _lib.ptr(gt_view), B, N, M, _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2),
                      _lib.ptr(i2), _lib.ptr(cws), cws.numel() * 4, 7, _lib.ptr(rec), K, Hr, Wr, _lib.ptr(order), s)
        else:
            _lib.call('vpn_chamfer_fwd_ws', _lib.ptr(points), _lib.ptr(gt_view), B, N, M, _lib.ptr(d1), _lib.ptr(i1), _lib.ptr(d2),
                      _lib.ptr(i2), _lib.ptr(cws), cws.numel() * 4, 7, s)
        if render:
            _lib.call('vpn_raster_total_fwd_fin', _lib.ptr(params), _lib.ptr(kinds), _lib.ptr(cam), B, K, H, W, float(sigma), float(gamma),
                      float(z_far), _lib.ptr(gt_sil), None, sil_mse, w_sil, 0.0, _lib.ptr(rec), _lib.ptr(lws), _lib.ptr(rws), 1,
                      _lib.ptr(cws), cws.numel() * 4, N, M, cd_w1, cd_w2, w_view, _lib.ptr(hot), None,
                      seed_dev if advance_seed else None, _lib.ptr(order), s)
        else:
            # train.py:167: no render; the Chamfer term alone (H = W = 0: no image losses)
            _lib.call('vpn_loss_finalize', _lib.ptr(lws), B, 0, 0, _lib.ptr(d1), _lib.ptr(d2), N, M, cd_w1, cd_w2, w_view, 0.0, 0.0,
                      _lib.ptr(hot), None, s)
            if advance_seed:
                raise ValueError('advance_seed rides in the raster launch: it needs the silhouette term (L_SIL != 0)')
        # ---- object-centred Chamfer (train.py:158-161): computed even at weight 0, like the reference
        cn = None
        if gt_canon is not None:
            gt_canon = _f32c(gt_canon)
            Mc = gt_canon.shape[1]
            cam4 = [_f32c(c.reshape(-1)) for c in (dists, elevs, azims, angles)]
            canon = torch.empty_like(points)
            _lib.call('vpn_camera_transform_fwd', _lib.ptr(points), _lib.ptr(cam4[0]), _lib.ptr(cam4[1]), _lib.ptr(cam4[2]),
                      _lib.ptr(cam4[3]), B, N, 1, _lib.ptr(canon), s)
            cd1, cd2 = torch.empty((B, N), **f32), torch.empty((B, Mc), **f32)
            ci1, ci2 = torch.empty((B, N), **i32), torch.empty((B, Mc), **i32)
            ccws = torch.empty((L.vpn_chamfer_workspace(B, N, Mc) // 4,), **f32)
            _lib.call('vpn_chamfer_fwd_ws', _lib.ptr(canon), _lib.ptr(gt_canon), B, N, Mc, _lib.ptr(cd1), _lib.ptr(ci1), _lib.ptr(cd2),
                      _lib.ptr(ci2), _lib.ptr(ccws), ccws.numel() * 4, 0, s)
            mat = None
            if w_can:
                mat = torch.empty((B, 9), **f32)
                _lib.call('vpn_camera_matrix', _lib.ptr(cam4[0]), _lib.ptr(cam4[1]), _lib.ptr(cam4[2]), _lib.ptr(cam4[3]), B, 1,
                          _lib.ptr(mat), s)
            cn = (canon, gt_canon, mat, cd1, ci1, cd2, ci2, Mc)
        # ---- VP-diversity (train.py:185)
        dv = dws = None
        if w_div:
            dv = (torch.empty((B, K), **f32), torch.empty((B, K), **i32), torch.empty((B, M), **f32), torch.empty((B, M), **i32))
            dws = torch.empty((L.vpn_vpdiv_workspace(B, K) // 8,), dtype=torch.int64, device=dev)
            # the centres' direction stays in the workspace: the finalisation's per-sample pass merges it
            _lib.call('vpn_vpdiv_fwd', _lib.ptr(params), _lib.ptr(gt_view), B, K, M, None, None, _lib.ptr(dv[2]),
                      _lib.ptr(dv[3]), _lib.ptr(dws), s)
        out = torch.empty((6,), **f32)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        fws = torch.empty((L.vpn_trainstep_workspace(B) // 4,), **f32)
        _lib.call('vpn_trainstep_finalize', _lib.ptr(hot), _lib.ptr(emd_dist), _lib.ptr(cn[3]) if cn else None,
                  _lib.ptr(cn[5]) if cn else None, None, _lib.ptr(dv[2]) if dv else None, B, N, M,
                  cn[7] if cn else 0, K, w_view, w_can, w_sil if render else 0.0, w_div, w_emd, cd_w1, cd_w2, _lib.ptr(fws),
                  _lib.ptr(dws) if dv else None, _lib.ptr(dv[0]) if dv else None, _lib.ptr(dv[1]) if dv else None, _lib.ptr(out), s)
        ctx.meta = (B, K, n, M, H, W, seed_host, int(sample_base), cd_w1, cd_w2, w_view, w_can, w_div, w_emd, render,
                    seed_dev is not None, cn[7] if cn else 0)
        # the sampler's launch keeps the seed it used at loss_ws + 8: backward reads it from there
        ctx.tensors = dict(params=params, kinds=kinds, cam=cam, gt_view=gt_view, points=points, d1=d1, i1=i1, d2=d2, i2=i2, rec=rec,
                           rws=rws, lws=lws, emd_dist=emd_dist, emd_assign=emd_assign, dv=dv, cn=cn if (cn and w_can) else None)
        terms = out.unbind(0)
        ctx.mark_non_differentiable(*terms[:5])
        ctx.set_materialize_grads(False)       # no zero-filled gradients for the five reported terms
        return terms

    @staticmethod
    def backward(ctx, _g0, _g1, _g2, _g3, _g4, grad_total):
        t = ctx.tensors
        if grad_total is None:
            return (None,) * 21
        (B, K, n, M, H, W, seed_host, base, cd_w1, cd_w2, w_view, w_can, w_div, w_emd, render, has_seed_dev, Mc) = ctx.meta
        N = K * n
        p = _lib.ptr
        g = _f32c(grad_total).reshape(1)
        grad_params = torch.empty_like(t['params'])
        seed_dev = ctypes.c_void_p(t['lws'].data_ptr() + 8) if has_seed_dev else None
        dv, cn = t['dv'], t['cn']
        _lib.call('vpn_trainstep_bwd', p(t['params']), p(t['kinds']), 0 if has_seed_dev else seed_host, seed_dev, base, B, K, n,
                  p(t['points']), p(t['gt_view']), M, p(t['d1']), p(t['i1']), p(t['d2']), p(t['i2']),
                  cd_w1 * w_view / B, cd_w2 * w_view / B, p(t['cam']), H, W, p(t['rec']) if render else None,
                  p(t['rws']) if render else None, p(g),
                  p(t['emd_dist']) if w_emd else None, p(t['emd_assign']) if w_emd else None, w_emd / (B * N),
                  p(dv[0]) if dv else None, p(dv[1]) if dv else None, p(dv[2]) if dv else None, p(dv[3]) if dv else None,
                  w_div * 0.5 / (K * B), w_div * 1.0 / (M * B),
                  p(cn[0]) if cn else None, p(cn[1]) if cn else None, p(cn[2]) if cn else None, p(cn[3]) if cn else None,
                  p(cn[4]) if cn else None, p(cn[5]) if cn else None, p(cn[6]) if cn else None,
                  (w_can * cd_w1 / (N * B)) if cn else 0.0, (w_can * cd_w2 / (Mc * B)) if cn else 0.0, Mc if cn else 0,
                  p(grad_params), _lib.stream())
        return (grad_params,) + (None,) * 20
